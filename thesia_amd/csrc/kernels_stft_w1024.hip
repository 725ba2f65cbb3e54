// kernels_stft_w1024.hip — the one-frame wave kernel's instantiations for n_fft 1024 (stft_wave_kernel<9, ...>) and their launcher:
// kernels_stft.hip compiled as its part 9 (see the note on translation units there).
#define TH_STFT_PART 9
#include "kernels_stft.hip"
