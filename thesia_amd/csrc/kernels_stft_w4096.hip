// kernels_stft_w4096.hip — the one-frame wave kernel's instantiations for n_fft 4096 (stft_wave_kernel<11, ...>) and their launcher:
// kernels_stft.hip compiled as its part 11 (see the note on translation units there).
#define TH_STFT_PART 11
#include "kernels_stft.hip"
