// conv + GroupNorm + GELU in one launch (conv_mfma_gn.inc), bf16_t instantiations
#include "conv_mfma_gn.inc"
int launch_conv_gn_bf16(const adk_op& op, hipStream_t st) { return launch_gn_taps<bf16_t>(op, st); }
