// Developer tool (round 4): the round-3 diffusion attention kernel (one tile body with three data paths and a per-tile branch), kept for the A/B in
// tools/attn_bench.hip against the product kernel (csrc/diffusion.hip: diff_attn_kernel, four tile ranges with one straight-line body each). Included after
// csrc/diffusion.hip, inside namespace tts. NOT part of the product.
#pragma once
namespace tts {
template <int NR> // K/V ring depth: 3 = two tiles in flight, 3 workgroups per CU; 2 = one tile in flight, 4 workgroups per CU
__global__ __launch_bounds__(256, NR == 2 ? 4 : 3) void diff_attn_r3_kernel(const __half *__restrict__ qk, const __half *__restrict__ vt, int ldvt,
                                                        const int *__restrict__ seq_start, const int *__restrict__ seq_len,
                                                        const float *__restrict__ bias_tab, __half *__restrict__ out, int nq) {
  // ONE LDS object: with a second __shared__ variable hipcc puts an s_waitcnt vmcnt(0) in front of the first
  // ds_read of every tile, which drains the DMA prefetch (seen in the ISA; cdna_hip_programming.md §5 trap (a)).
  // (dynamic LDS: with a static array the DMA writes and the fragment reads alias for the waitcnt pass as well)
  extern __shared__ __attribute__((aligned(16))) char smem[]; // 3 x (K tile 8 KB | V^T tile 8 KB) + bias table
  float *tab = (float *)(smem + NR * 16384);
  // XCD-aware block order: workgroup id b runs on XCD b % 8, so all q-blocks of one (sequence, head) pair
  // get ids congruent mod 8 and reuse that pair's K/V tiles from one L2 (16 heads => pairs % 8 == 0).
  const int xcd = blockIdx.x & 7, tt = blockIdx.x >> 3;
  const int pair = (tt / nq) * 8 + xcd, h = pair & 15, s = pair >> 4;
  const int T = seq_len[s], r0 = seq_start[s], q0 = (tt % nq) * 128;
  if (q0 >= T) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), fr = lane & 15, fq = lane >> 4;
  const float L2E = 1.44269504088896f;
  // Bias by SIGNED key-query distance d in [-160, 160), saturated outside +-63, in raw-score units (added to q.k
  // before the 1/8 * log2e scaling): a lane's 16 keys of a tile sit at compile-time offsets from one base distance,
  // so the near-diagonal path is one LDS read at an immediate offset + one add per score.
  const float SC = 0.125f * L2E; // 1/sqrt(64) in log2 units (softmax via exp2)
  for (int j = tid; j < ATT_TAB; j += 256) {
    const int d = j - ATT_TAB / 2, ad = d < 0 ? -d : d;
    tab[j] = bias_tab[h * 128 + (d > 0 ? 64 : 0) + (ad < 63 ? ad : 63)] * (L2E / SC);
  }
  const int qw = q0 + wave * 32;
  half8 qf[2][2]; // Q[query = qw + i*16 + fr][d = ks*32 + fq*8 ..+7]
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int ks = 0; ks < 2; ks++)
      qf[i][ks] = *(const half8 *)(qk + (size_t)(r0 + qw + i * 16 + fr) * 2048 + h * 128 + ks * 32 + fq * 8);
  // Retire the Q loads HERE (a use makes hipcc place its vmcnt(0) now): vmcnt is an in-order counter, so a Q
  // load still pending at the loop would force vmcnt(0) in front of the first MFMA of every tile and drain the
  // K/V prefetch (seen in the ISA as `s_waitcnt vmcnt(0) lgkmcnt(0)` after the ds_reads).
  asm volatile("" ::"v"(qf[0][0]), "v"(qf[0][1]), "v"(qf[1][0]), "v"(qf[1][1]));
  floatx4 o[2][4]; // O^T[d = dt*16 + fq*4 + r][query = qw + i*16 + fr]
  floatx4 lacc[2]; // row sums of P from the matrix pipe: (all-ones A tile) . P^T, every register = l[query fr]
  float mrow[2];
#pragma unroll
  for (int i = 0; i < 2; i++) {
#pragma unroll
    for (int j = 0; j < 4; j++) o[i][j] = (floatx4){0.f, 0.f, 0.f, 0.f};
    lacc[i] = (floatx4){0.f, 0.f, 0.f, 0.f};
    mrow[i] = -INFINITY;
  }
  const int nkb = (T + 63) >> 6;
  const int prow = lane >> 3, pslot = lane & 7;
  const __half *kbase = qk + (size_t)r0 * 2048 + h * 128 + 64;
  const __half *vbase = vt + (size_t)(h * 64) * ldvt + r0;
  // K and V^T tiles live in a 3-deep ring of (K 8 KB | V^T 8 KB) slots filled by LDS-DMA two tiles ahead.
  // Wave w moves rows w*16 .. w*16+15 of both tiles; swizzle on the source chunk.
  // The K tile is stored with its key rows permuted: LDS row jt*16 + x holds key SIG(jt, x) =
  // (jt>>1)*32 + (x>>2)*8 + (jt&1)*4 + (x&3). The score accumulator (jt, fq, r) then belongs to key
  // (jt>>1)*32 + fq*8 + (jt&1)*4 + r, so the 8 P values a lane feeds to PV step ks2 are the 8 CONSECUTIVE keys
  // 32 ks2 + 8 fq .. +7 and its V^T fragment is one 16-byte LDS read (no half-fragment shuffles).
  // Tile indices past the end are clamped (harmless re-stage) so that the vmcnt arithmetic stays uniform.
  int koff[2], voff[2]; // per-lane source offsets (halves) inside a tile, fixed for the whole kernel
#pragma unroll
  for (int i = 0; i < 2; i++) {
    const int row = wave * 16 + i * 8 + prow, c = pslot ^ (row & 7);
    const int jt = row >> 4, x = row & 15;
    const int key = (jt >> 1) * 32 + (x >> 2) * 8 + (jt & 1) * 4 + (x & 3);
    koff[i] = key * 2048 + c * 8;
    voff[i] = row * ldvt + c * 8;
  }
  auto stage = [&](int kb, int slot) {
    kb = min(kb, nkb - 1);
    const __half *ksrc = kbase + (size_t)kb * (64 * 2048), *vsrc = vbase + kb * 64; // wave-uniform
    char *ks_ = smem + slot * 16384 + wave * 2048, *vs_ = ks_ + 8192;
    __builtin_amdgcn_global_load_lds((gptr_t)(ksrc + koff[0]), (lptr_t)ks_, 16, 0, 0);
    __builtin_amdgcn_global_load_lds((gptr_t)(ksrc + koff[1]), (lptr_t)(ks_ + 1024), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((gptr_t)(vsrc + voff[0]), (lptr_t)vs_, 16, 0, 0);
    __builtin_amdgcn_global_load_lds((gptr_t)(vsrc + voff[1]), (lptr_t)(vs_ + 1024), 16, 0, 0);
  };
  // S^T of one key tile: sc[i][jt][r] = S[query i*16+fr][key SIG(jt, fq*4 + r)]
  auto scores = [&](const char *Ks, floatx4 (&sc)[2][4]) {
#pragma unroll
    for (int jt = 0; jt < 4; jt++) {
      const half8 kf0 = *(const half8 *)(Ks + attn_off(jt * 16 + fr, fq));
      const half8 kf1 = *(const half8 *)(Ks + attn_off(jt * 16 + fr, 4 + fq));
#pragma unroll
      for (int i = 0; i < 2; i++) {
        floatx4 a = (floatx4){0.f, 0.f, 0.f, 0.f};
        a = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf0, qf[i][0], a, 0, 0, 0);
        a = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf1, qf[i][1], a, 0, 0, 0);
        sc[i][jt] = a;
      }
    }
  };
  ATT_CLK(0);
  stage(0, 0);
  if (NR == 3) stage(1, 1);
  half8 ones;
#pragma unroll
  for (int e = 0; e < 8; e++) ones[e] = (_Float16)1.0f;
  for (int kb = 0; kb < nkb; kb++) {
    // Tile kb must have landed; the 4 DMA pieces of tile kb+1 may stay in flight across the barrier
    // (counted vmcnt + raw s_barrier: __syncthreads() would drain the prefetch).
    ATT_T(0);
    if (NR == 3) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    ATT_T(1);
    __builtin_amdgcn_s_barrier();
    ATT_T(2);
    // every wave has passed the barrier => nobody still reads tile kb-1, whose slot receives the next tile to stage
    stage(kb + NR - 1, (kb + NR - 1) % NR);
    const char *Ks = smem + (kb % NR) * 16384, *Vs = Ks + 8192;
    ATT_T(3);
    floatx4 sc[2][4];
    scores(Ks, sc);
    ATT_T(4);
    const int kmin = kb * 64;
    const bool far_hi = kmin - (qw + 31) >= 63, far_lo = qw - (kmin + 63) >= 63, tail = kmin + 64 > T;
    half8 pf[2][2]; // P^T in B-operand layout: slot e of step ks2 = key 32 ks2 + 8 fq + e
#pragma unroll
    for (int i = 0; i < 2; i++) {
      const int qi = qw + i * 16 + fr;
      float mx = -INFINITY, boff = 0.f; // v = sc*SC + bias; far tiles: bias is one constant (folded below)
      if ((far_hi || far_lo) && !tail) {
        boff = (far_hi ? tab[ATT_TAB / 2 + 63] : tab[ATT_TAB / 2 - 63]) * SC;
#pragma unroll
        for (int jt = 0; jt < 4; jt++) { // two v_max3 per accumulator register quad
          mx = fmaxf(fmaxf(mx, sc[i][jt][0]), sc[i][jt][1]);
          mx = fmaxf(fmaxf(mx, sc[i][jt][2]), sc[i][jt][3]);
        }
        mx = fmaf(mx, SC, boff);
      } else if (!tail) {
        // key of (jt, r) = kmin + fq*8 + off, off = (jt>>1)*32 + (jt&1)*4 + r  =>  d = (kmin + fq*8 - qi) + off
        const float *tp = tab + (kmin + fq * 8 - qi + ATT_TAB / 2); // in range: |d| < 160 on near-diagonal tiles
#pragma unroll
        for (int jt = 0; jt < 4; jt++)
#pragma unroll
          for (int r = 0; r < 4; r++) {
            const float v = sc[i][jt][r] + tp[(jt >> 1) * 32 + (jt & 1) * 4 + r];
            sc[i][jt][r] = v;
            mx = fmaxf(mx, v);
          }
        mx *= SC;
      } else { // last tile of the sequence: keys >= T are masked
        const int left = T - kmin - fq * 8; // keys of this lane with off < left exist
        const bool far = far_hi || far_lo;  // then the bias is one constant (and the table base would be out of range)
        const float cb = far_hi ? tab[ATT_TAB / 2 + 63] : tab[ATT_TAB / 2 - 63];
        const float *tp = tab + (far ? 0 : kmin + fq * 8 - qi + ATT_TAB / 2);
        float bv[4][4]; // all table reads first, unconditionally (a load under a per-element select is branched around)
#pragma unroll
        for (int jt = 0; jt < 4; jt++)
#pragma unroll
          for (int r = 0; r < 4; r++) bv[jt][r] = tp[(jt >> 1) * 32 + (jt & 1) * 4 + r];
#pragma unroll
        for (int jt = 0; jt < 4; jt++)
#pragma unroll
          for (int r = 0; r < 4; r++) {
            const int off = (jt >> 1) * 32 + (jt & 1) * 4 + r;
            const float v = off < left ? sc[i][jt][r] + (far ? cb : bv[jt][r]) : -INFINITY;
            sc[i][jt][r] = v;
            mx = fmaxf(mx, v);
          }
        mx *= SC;
      }
      mx = rows4_max(mx);
      const float mnew = fmaxf(mrow[i], mx);
      const float alpha = __builtin_amdgcn_exp2f(mrow[i] - mnew);
      const float sub = boff - mnew; // p = 2^(sc*SC + bias - mnew)
      mrow[i] = mnew;
      if (!__all(alpha == 1.0f)) { // the running max settles after the first tiles: usually nothing to rescale
#pragma unroll
        for (int dt = 0; dt < 4; dt++)
#pragma unroll
          for (int r = 0; r < 4; r++) o[i][dt][r] *= alpha;
#pragma unroll
        for (int r = 0; r < 4; r++) lacc[i][r] *= alpha;
      }
#pragma unroll
      for (int ks2 = 0; ks2 < 2; ks2++)
#pragma unroll
        for (int e = 0; e < 8; e++)
          pf[i][ks2][e] = (_Float16)__builtin_amdgcn_exp2f(fmaf(sc[i][2 * ks2 + (e >> 2)][e & 3], SC, sub));
    }
    ATT_T(5);
    // O^T += V^T P^T : A = V^T[d = dt*16 + fr][keys 32 ks2 + 8 fq ..+7], B = P^T; row sums: A = ones
#pragma unroll
    for (int ks2 = 0; ks2 < 2; ks2++) {
#pragma unroll
      for (int dt = 0; dt < 4; dt++) {
        const half8 vf = *(const half8 *)(Vs + attn_off(dt * 16 + fr, 4 * ks2 + fq));
#pragma unroll
        for (int i = 0; i < 2; i++) o[i][dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf, pf[i][ks2], o[i][dt], 0, 0, 0);
      }
#pragma unroll
      for (int i = 0; i < 2; i++) lacc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ones, pf[i][ks2], lacc[i], 0, 0, 0);
    }
    ATT_T(6);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // trailing (clamped) DMA pieces must land before the LDS is released
  ATT_CLK(1);
  float lrow[2] = {lacc[0][0], lacc[1][0]};
#pragma unroll
  for (int i = 0; i < 2; i++) {
    const int qi = qw + i * 16 + fr;
    if (qi < T) {
      const float inv = 1.0f / lrow[i];
#pragma unroll
      for (int dt = 0; dt < 4; dt++) {
        __half2 p0 = __floats2half2_rn(o[i][dt][0] * inv, o[i][dt][1] * inv), p1 = __floats2half2_rn(o[i][dt][2] * inv, o[i][dt][3] * inv);
        uint2 u;
        u.x = *(unsigned *)&p0;
        u.y = *(unsigned *)&p1;
        *(uint2 *)(out + (size_t)(r0 + qi) * C + h * 64 + dt * 16 + fq * 4) = u;
      }
    }
  }
}

} // namespace tts
