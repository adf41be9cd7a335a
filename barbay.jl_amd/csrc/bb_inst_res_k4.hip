// bb_inst_res_k4.hip -- the k_res instances of model kind 4 (bb_resident.h), one translation unit of the library (see bb_inst.h)
#include "bb_inst.h"
#define BR_K 4
#define BR_T_1024 
#define BR_P_1024 BR_CASE(1, 1024)
#define BR_T_512 BR_T(3, 512, 6)
#include "bb_inst_res.inc"
