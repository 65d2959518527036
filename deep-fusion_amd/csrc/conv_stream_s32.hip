// conv_stream_s32.hip -- s32-output instantiations of the streamed-weight MFMA conv kernel.
#define DFX_INST_DST DFX_S32
#define DFX_INST_NAME launch_conv_stream_s32
#include "conv_stream_inst.inc"
