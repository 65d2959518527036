// conv_direct_s8.hip -- s8-output instantiations of the direct-weight MFMA fused conv kernel.
#define DFX_INST_DST DFX_S8
#define DFX_INST_NAME launch_conv_direct_s8
#include "conv_direct_inst.inc"
