// conv_stream_f32.hip -- f32-output instantiations of the streamed-weight MFMA conv kernel.
#define DFX_INST_DST DFX_F32
#define DFX_INST_NAME launch_conv_stream_f32
#include "conv_stream_inst.inc"
