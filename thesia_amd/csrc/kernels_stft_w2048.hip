// kernels_stft_w2048.hip — the one-frame wave kernel's instantiations for n_fft 2048 (stft_wave_kernel<10, ...>) and their launcher:
// kernels_stft.hip compiled as its part 10 (see the note on translation units there).
#define TH_STFT_PART 10
#include "kernels_stft.hip"
