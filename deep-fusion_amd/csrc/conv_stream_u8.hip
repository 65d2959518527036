// conv_stream_u8.hip -- u8-output instantiations of the streamed-weight MFMA conv kernel.
#define DFX_INST_DST DFX_U8
#define DFX_INST_NAME launch_conv_stream_u8
#include "conv_stream_inst.inc"
