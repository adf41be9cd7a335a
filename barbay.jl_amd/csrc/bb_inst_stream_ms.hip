// bb_inst_stream_ms.hip -- the k_stream instances that take several MC samples per step and / or record the ELBO (bb_stream.h, MS = true),
// one translation unit of the library (see bb_inst.h)
#include "bb_inst.h"
bb_stream_kernel bb_stream_instance_ms(int kind, int nthr, int T, const char** nm) {
    if (nm) *nm = "";
#define BS_CASE(K, NT, TT) if (kind == (K) && nthr == (NT) && T == (TT)) { if (nm) *nm = "k_stream<" #K "," #NT "," #TT ",true>"; return k_stream<K, NT, TT, true>; }
    BS_CASE(0, 1024, 8) BS_CASE(1, 1024, 8) BS_CASE(2, 1024, 8) BS_CASE(3, 1024, 8) BS_CASE(4, 1024, 8)
    BS_CASE(0, 1024, 6) BS_CASE(1, 1024, 6) BS_CASE(2, 1024, 6) BS_CASE(3, 1024, 6) BS_CASE(4, 1024, 6)
    BS_CASE(2, 512, 8) BS_CASE(3, 512, 6)
#undef BS_CASE
    return nullptr;
}
