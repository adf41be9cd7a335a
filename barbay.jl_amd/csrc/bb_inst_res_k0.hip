// bb_inst_res_k0.hip -- the k_res instances of model kind 0 (bb_resident.h), one translation unit of the library (see bb_inst.h)
#include "bb_inst.h"
#define BR_K 0
#define BR_T_1024 BR_T(1, 1024, 8) BR_T(1, 1024, 6)
#define BR_P_1024 BR_CASE(1, 1024) BR_CASE(2, 1024)
#define BR_T_512 BR_T(1, 512, 8) BR_T(2, 512, 8)
#include "bb_inst_res.inc"
