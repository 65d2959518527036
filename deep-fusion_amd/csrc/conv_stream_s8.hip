// conv_stream_s8.hip -- s8-output instantiations of the streamed-weight MFMA conv kernel.
#define DFX_INST_DST DFX_S8
#define DFX_INST_NAME launch_conv_stream_s8
#include "conv_stream_inst.inc"
