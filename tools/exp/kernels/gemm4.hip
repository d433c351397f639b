// tile_hint 7: the 256 x 256 LDS-DMA kernel of gemm2.hip instantiated with FOUR waves (one per SIMD), wave tile 64 x 256.
// A wave then re-reads (4 + 16) fragments per 32-deep step for 64 MFMAs: 160 KiB of LDS reads per K step per CU against
// 256 KiB in the 16-wave form (whose 64 x 64 wave tiles need exactly the 128 B/clk the LDS can deliver at full MFMA rate).
// The 256 accumulator registers per lane live in the AGPR file, so this translation unit is built WITHOUT
// -amdgpu-mfma-vgpr-form (see build.py).
#define G2_NS g4
#define G2_VARIANT_ONLY 1
#include "gemm2.hip"

int s4f_gemm4_try(const s4f_gemm_desc& d, hipStream_t st) {
  if (d.dtype != S4F_BF16) return -100;
  if (d.b_mode == S4F_OP_K_CONV && (d.cC % 256) != 0) return -100;
  return g4::dispatch<256, 4>(d, st);
}
