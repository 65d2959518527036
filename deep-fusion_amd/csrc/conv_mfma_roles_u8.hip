// conv_mfma_roles_u8.hip -- u8-output instantiations of the role-specialised fused MFMA conv kernel.
#define DFX_INST_DST DFX_U8
#define DFX_INST_NAME launch_conv_mfma_roles_u8
#include "conv_mfma_roles_inst.inc"
