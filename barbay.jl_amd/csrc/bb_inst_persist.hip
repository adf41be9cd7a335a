// bb_inst_persist.hip -- the k_persist instances (bb_persist.h), one translation unit of the library (see bb_inst.h)
#include "bb_inst.h"

bb_persist_kernel bb_persist_instance(int kind, int P, int nthr, bool xg, const char** nm) {
    if (nm) *nm = "";
    if (xg) {                  // sharded tiles are small: one pair per thread only
        if (P != 1) return nullptr;
        if (nthr > 512) switch (kind) {
            case 0: { if (nm) *nm = "k_persist<0,1,1024,true>"; return k_persist<0, 1, 1024, true>; }  case 1: { if (nm) *nm = "k_persist<1,1,1024,true>"; return k_persist<1, 1, 1024, true>; }
            case 3: { if (nm) *nm = "k_persist<3,1,1024,true>"; return k_persist<3, 1, 1024, true>; }  case 4: { if (nm) *nm = "k_persist<4,1,1024,true>"; return k_persist<4, 1, 1024, true>; }
            default: return nullptr;
        }
        switch (kind) {
        case 0: { if (nm) *nm = "k_persist<0,1,512,true>"; return k_persist<0, 1, 512, true>; }  case 1: { if (nm) *nm = "k_persist<1,1,512,true>"; return k_persist<1, 1, 512, true>; }
        case 3: { if (nm) *nm = "k_persist<3,1,512,true>"; return k_persist<3, 1, 512, true>; }  case 4: { if (nm) *nm = "k_persist<4,1,512,true>"; return k_persist<4, 1, 512, true>; }
        default: return nullptr;
        }
    }
    if (nthr > 512) {          // 16 waves per CU: 128 VGPRs per lane
        switch (kind * 10 + P) {
        case 1: { if (nm) *nm = "k_persist<0,1,1024>"; return k_persist<0, 1, 1024>; }   case 2: { if (nm) *nm = "k_persist<0,2,1024>"; return k_persist<0, 2, 1024>; }
        case 11: { if (nm) *nm = "k_persist<1,1,1024>"; return k_persist<1, 1, 1024>; }  case 12: { if (nm) *nm = "k_persist<1,2,1024>"; return k_persist<1, 2, 1024>; }
        case 31: { if (nm) *nm = "k_persist<3,1,1024>"; return k_persist<3, 1, 1024>; }  case 32: { if (nm) *nm = "k_persist<3,2,1024>"; return k_persist<3, 2, 1024>; }
        case 41: { if (nm) *nm = "k_persist<4,1,1024>"; return k_persist<4, 1, 1024>; }  case 42: { if (nm) *nm = "k_persist<4,2,1024>"; return k_persist<4, 2, 1024>; }
        default: return nullptr;
        }
    }
    switch (kind * 10 + P) {   // <= 8 waves per CU: 256 VGPRs per lane, more pairs per thread
    case 1: { if (nm) *nm = "k_persist<0,1,512>"; return k_persist<0, 1, 512>; }   case 2: { if (nm) *nm = "k_persist<0,2,512>"; return k_persist<0, 2, 512>; }   case 3: { if (nm) *nm = "k_persist<0,3,512>"; return k_persist<0, 3, 512>; }   case 4: { if (nm) *nm = "k_persist<0,4,512>"; return k_persist<0, 4, 512>; }
    case 11: { if (nm) *nm = "k_persist<1,1,512>"; return k_persist<1, 1, 512>; }  case 12: { if (nm) *nm = "k_persist<1,2,512>"; return k_persist<1, 2, 512>; }  case 13: { if (nm) *nm = "k_persist<1,3,512>"; return k_persist<1, 3, 512>; }  case 14: { if (nm) *nm = "k_persist<1,4,512>"; return k_persist<1, 4, 512>; }
    case 31: { if (nm) *nm = "k_persist<3,1,512>"; return k_persist<3, 1, 512>; }  case 32: { if (nm) *nm = "k_persist<3,2,512>"; return k_persist<3, 2, 512>; }  case 33: { if (nm) *nm = "k_persist<3,3,512>"; return k_persist<3, 3, 512>; }  case 34: { if (nm) *nm = "k_persist<3,4,512>"; return k_persist<3, 4, 512>; }
    case 41: { if (nm) *nm = "k_persist<4,1,512>"; return k_persist<4, 1, 512>; }  case 42: { if (nm) *nm = "k_persist<4,2,512>"; return k_persist<4, 2, 512>; }  case 43: { if (nm) *nm = "k_persist<4,3,512>"; return k_persist<4, 3, 512>; }  case 44: { if (nm) *nm = "k_persist<4,4,512>"; return k_persist<4, 4, 512>; }
    default: return nullptr;
    }
}
