// conv_direct_f32.hip -- f32-output instantiations of the direct-weight MFMA fused conv kernel.
#define DFX_INST_DST DFX_F32
#define DFX_INST_NAME launch_conv_direct_f32
#include "conv_direct_inst.inc"
