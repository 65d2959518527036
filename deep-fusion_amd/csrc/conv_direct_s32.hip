// conv_direct_s32.hip -- s32-output instantiations of the direct-weight MFMA fused conv kernel.
#define DFX_INST_DST DFX_S32
#define DFX_INST_NAME launch_conv_direct_s32
#include "conv_direct_inst.inc"
