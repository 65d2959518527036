// conv_direct_u8.hip -- u8-output instantiations of the direct-weight MFMA fused conv kernel.
#define DFX_INST_DST DFX_U8
#define DFX_INST_NAME launch_conv_direct_u8
#include "conv_direct_inst.inc"
