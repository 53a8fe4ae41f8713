// PROBE (round 5; not compiled into libvtc_hip.so): the sweep's distance GEMM + block-minima epilogue on TWO workgroups per CU.
// To build it, paste the kernel into vtc_amd/csrc/gemm.hip's anonymous namespace behind gemm_phased_kernel (it uses l2min_epilogue,
// glds16s, lds_read16, Mma, GemmParams) and the host part in front of run_l2min (commit 3-a of round 5 had it under VTC_SWEEP_GEMM=2).
// Result: all EXACT / bidir / sharded sweep tests pass (ids identical); 10k 0.51 ms, 50k 5.2 ms against 0.49 / 4.6 ms of the phased
// 256 x 256 kernel, whatever the start stagger (0 - 8 us): a 128 x 256 tile needs 1.5 x the operand bytes per flop, and with 72 KiB of
// LDS per workgroup only two K = 32 slabs (2 x 24 KiB) are in flight -- the K loop runs DMA-latency-bound (~0.8 us per K-step where the
// matrix work of a step is 0.25 us), so the partner's epilogue hides under a K loop that is itself 2-3 x too long.

// =====================================================================================================
// EPI_L2MIN on TWO workgroups per CU (round 5).  The block-minima epilogue is ~2 800 vector instructions per wave and tile --
// as long as the tile's K loop (K = 512) -- and in the phased kernel all eight waves of the CU's one workgroup run it at the
// same time with the matrix pipe idle.  Here a workgroup is FOUR waves (one per SIMD) on a 128 x 256 tile (the same 128 x 64
// wave tile, the same row blocks of 128, the same key planes), two workgroups share a CU, and the second one starts half a
// tile late: one workgroup's epilogue (vector ALU) runs under the other's K loop (matrix pipe, LDS, LDS-DMA).  Each gets
// 72 KiB of LDS: three stages of a K = 32 slab (64-byte rows: 128 + 256 rows = 24 KiB), two slabs in flight, one barrier per
// K-step; whatever latency a K-step still exposes is the partner's to fill.
// LDS image: rows of 64 B = four 16-byte chunks, lane-linear for the LDS-DMA (a piece = 16 rows); chunk slot = chunk ^ ((row >> 2) & 3),
// applied on the DMA source address and on the ds_read_b128 address: the 16 rows of a fragment read hit 16 distinct bank quadruples.
__global__ __launch_bounds__(256, 2) void gemm_l2min2_kernel(GemmParams p) {
  constexpr int WM = 1, WN = 4, TM = 8, TN = 4, BM = 128, BN = 256, RB = 64, NST = 3;
  constexpr int A_BYTES = BM * RB, STAGE = (BM + BN) * RB;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, l15 = lane & 15;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);

  // persistent, XCD-aware walk over 128 x 256 tiles (as gemm_kernel: super-rows of SUPER_ROWS rows, column by column)
  constexpr int SUPER = SUPER_ROWS / BM;
  const int ntiles = p.MT * p.NT, nwg = gridDim.x, bid = blockIdx.x;
  const int xcd = bid & 7, slot = bid >> 3;
  const int nb_x = (nwg >> 3) + (xcd < (nwg & 7) ? 1 : 0);
  const int nt_x = (ntiles >> 3) + (xcd < (ntiles & 7) ? 1 : 0);
  const int start_x = xcd * (ntiles >> 3) + min(xcd, ntiles & 7);
  auto decode = [&](int logical, int &m0, int &n0) {
    const int per_super = SUPER * p.NT;
    const int sr = logical / per_super, rem = logical - sr * per_super;
    const int gsz = min(SUPER, p.MT - sr * SUPER);
    const int nt = rem / gsz;
    m0 = (sr * SUPER + (rem - nt * gsz)) * BM;
    n0 = nt * BN;
  };
  if (slot >= nt_x) return;                     // uniform for the workgroup

  // The workgroup whose LDS allocation does not start at 0 is the second one on its CU: it starts `stagger_ticks` x 10 ns late
  // (HW_REG_LDS_ALLOC, bits [11:0] = base; if the field reads 0 for both the two stay in step -- slower, never wrong).
  if (p.stagger_ticks > 0) {
    const unsigned lds_alloc_base = __builtin_amdgcn_s_getreg((11 << 11) | (0 << 6) | 6);
    if (lds_alloc_base != 0) {
      const long long t_end = (long long)__builtin_amdgcn_s_memrealtime() + (long long)p.stagger_ticks;
      while ((long long)__builtin_amdgcn_s_memrealtime() < t_end) __builtin_amdgcn_s_sleep(8);
    }
  }

  const int ksteps = p.K / 32;                  // host: K % 32 == 0, K >= 64
  const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_void *)lds);
  // fragment read offsets inside a stage
  const unsigned rd_lane = (unsigned)l15 * RB + (unsigned)((g ^ ((l15 >> 2) & 3)) << 4);
  const unsigned a_rd = rd_lane, w_rd = A_BYTES + (unsigned)(wave * 64) * RB + rd_lane;
  // LDS-DMA: lane -> (row lane >> 2 of the 16-row piece, chunk slot lane & 3); source chunk = slot ^ ((row >> 2) & 3)
  const unsigned src_chunk = (unsigned)(((lane & 3) ^ ((lane >> 4) & 3)) << 4);

  for (int li = slot; li < nt_x; li += nb_x) {
    int m0, n0;
    decode(start_x + li, m0, n0);
    unsigned a_off[2], w_off[4];
#pragma unroll
    for (int q = 0; q < 2; ++q) a_off[q] = (unsigned)min(m0 + (2 * wave_u + q) * 16 + (lane >> 2), p.M - 1) * (unsigned)p.lda_bytes + src_chunk;
#pragma unroll
    for (int q = 0; q < 4; ++q) w_off[q] = (unsigned)min(n0 + (4 * wave_u + q) * 16 + (lane >> 2), p.N - 1) * (unsigned)p.ldw_bytes + src_chunk;
    auto issue = [&](int k, int st) __attribute__((always_inline)) {          // 6 LDS-DMA pieces per wave, always
      const unsigned dst = lds_base + (unsigned)st * STAGE;
#pragma unroll
      for (int q = 0; q < 2; ++q) glds16s(a_off[q], p.A + (size_t)k * RB, __builtin_amdgcn_readfirstlane(dst + (2 * wave_u + q) * 1024));
#pragma unroll
      for (int q = 0; q < 4; ++q) glds16s(w_off[q], p.W + (size_t)k * RB, __builtin_amdgcn_readfirstlane(dst + A_BYTES + (4 * wave_u + q) * 1024));
    };
    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    issue(0, 0);
    issue(1, 1);
    int st_rd = 0, st_wr = 2;                    // stage read at step t, stage filled with step t + 2
    for (int t = 0; t < ksteps; ++t) {
      // my pieces of step t have landed (the queue is in order: the previous tile's stores, step t, step t + 1) ...
      if (t + 1 < ksteps) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();              // ... everybody's have, and everybody is past its reads of step t - 1
      asm volatile("" ::: "memory");
      if (t + 2 < ksteps) issue(t + 2, st_wr);   // into the stage step t - 1 was read from
      const unsigned sb = lds_base + (unsigned)st_rd * STAGE;
      u32x4 wf[TN], af[TM];
#pragma unroll
      for (int j = 0; j < TN; ++j) lds_read16(wf[j], sb + w_rd, j * 16 * RB);
#pragma unroll
      for (int i = 0; i < TM; ++i) lds_read16(af[i], sb + a_rd, i * 16 * RB);
      asm volatile("s_waitcnt lgkmcnt(0)"
                   : "+v"(wf[0]), "+v"(wf[1]), "+v"(wf[2]), "+v"(wf[3]), "+v"(af[0]), "+v"(af[1]), "+v"(af[2]), "+v"(af[3]), "+v"(af[4]),
                     "+v"(af[5]), "+v"(af[6]), "+v"(af[7]));
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) Mma<bf16_t>::run(wf[j], af[i], acc[i][j]);
      st_rd = st_rd == NST - 1 ? 0 : st_rd + 1;
      st_wr = st_wr == NST - 1 ? 0 : st_wr + 1;
    }
    __builtin_amdgcn_s_barrier();                // every wave is past its last fragment read before the next tile's first pieces land
    asm volatile("" ::: "memory");
    l2min_epilogue<WM, WN, TM, TN>(acc, p, m0, n0);
  }
}


// ---- host side ----
int run_l2min2(GemmParams p, hipStream_t stream) {
  p.MT = cdiv(p.M, 128); p.NT = cdiv(p.N, 256);
  const int ntiles = p.MT * p.NT;
  constexpr int shmem = 3 * (128 + 256) * 64;            // 72 KiB: two workgroups per CU
  static PerDeviceOnce attr;
  if (ensure_dynamic_lds(attr, reinterpret_cast<const void *>(&gemm_l2min2_kernel), shmem, "gemm_l2min2")) return 1;
  static const int ticks = [] { const char *e = getenv("VTC_SWEEP_STAGGER"); return e ? atoi(e) : 500; }();   // x 10 ns: half a tile
  p.stagger_ticks = ticks;
  hipLaunchKernelGGL(gemm_l2min2_kernel, dim3(min(ntiles, 2 * num_cus())), dim3(256), shmem, stream, p);
  VTC_LAUNCH_CHECK("gemm_l2min2");
  return 0;
}
// 0 = the phased 256 x 256 kernel (one workgroup per CU), 2 = two 4-wave workgroups per CU on 128 x 256 tiles (VTC_SWEEP_GEMM)
int l2min_gemm_variant() {
  static const int v = [] { const char *e = getenv("VTC_SWEEP_GEMM"); return e ? atoi(e) : 0; }();
  return v;
}
