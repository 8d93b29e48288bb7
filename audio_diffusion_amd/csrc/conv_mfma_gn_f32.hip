// conv + GroupNorm + GELU in one launch (conv_mfma_gn.inc), float instantiations
#include "conv_mfma_gn.inc"
int launch_conv_gn_f32(const adk_op& op, hipStream_t st) { return launch_gn_taps<float>(op, st); }
