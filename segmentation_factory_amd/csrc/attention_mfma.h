// Internal interface between attention.hip (C-ABI entry points, fp32 VALU kernels) and attention_mfma.hip (bf16 MFMA kernels).
#pragma once
#include "common.h"

// all pointers bf16; rows 16-byte aligned (checked by the caller); hd in {32, 64}
int attn_mfma_fwd(int hd, int B, int heads, int N, int Nkv, const void* q, int64_t ldq, const void* k, int64_t ldk,
                  const void* v, int64_t ldv, float scale, void* o, int64_t ldo, float* lse, hipStream_t st);
// Dbuf: [B][heads][N] floats; slab: [nchunk][B*Nkv][2*heads*hd] floats (dk columns then dv columns), query chunk z covers
// queries [z*qchunk, (z+1)*qchunk)
int attn_mfma_bwd(int hd, int B, int heads, int N, int Nkv, const void* q, int64_t ldq, const void* k, int64_t ldk,
                  const void* v, int64_t ldv, float scale, const void* o, int64_t ldo, const void* d_o, int64_t lddo,
                  const float* lse, void* dq, int64_t lddq, float* Dbuf, float* slab, int nchunk, int qchunk, hipStream_t st);
