// conv_mfma_roles_s8.hip -- s8-output instantiations of the role-specialised fused MFMA conv kernel.
#define DFX_INST_DST DFX_S8
#define DFX_INST_NAME launch_conv_mfma_roles_s8
#include "conv_mfma_roles_inst.inc"
