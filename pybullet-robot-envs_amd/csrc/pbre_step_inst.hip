// pbre_step_inst.hip -- one explicit instantiation of the Panda step launcher (pbre_panda.hpp), compiled once per (MODE, RT):
//   hipcc ... -DPBRE_INST_MODE=<0..5> -DPBRE_INST_RT=<false|true> -c -o obj/pbre_step_<m>_<rt>.o pbre_step_inst.hip        (build.sh)
#include "pbre_panda.hpp"
PBRE_STEP_INST(, PBRE_INST_MODE, PBRE_INST_RT)
