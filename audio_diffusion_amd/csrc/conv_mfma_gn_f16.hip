// conv + GroupNorm + GELU in one launch (conv_mfma_gn.inc), f16_t instantiations
#include "conv_mfma_gn.inc"
int launch_conv_gn_f16(const adk_op& op, hipStream_t st) { return launch_gn_taps<f16_t>(op, st); }
