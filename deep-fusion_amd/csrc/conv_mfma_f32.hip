// conv_mfma_f32.hip -- f32-output instantiations of the fused MFMA conv kernel.
#define DFX_INST_DST DFX_F32
#define DFX_INST_NAME launch_conv_mfma_f32
#include "conv_mfma_inst.inc"
