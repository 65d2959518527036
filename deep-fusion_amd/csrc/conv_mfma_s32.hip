// conv_mfma_s32.hip -- s32-output instantiations of the fused MFMA conv kernel.
#define DFX_INST_DST DFX_S32
#define DFX_INST_NAME launch_conv_mfma_s32
#include "conv_mfma_inst.inc"
