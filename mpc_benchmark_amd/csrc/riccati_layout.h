// riccati_layout.h — the LDS plan of the Riccati sweep (riccati_mfma.h) as plain constexpr host / device code: which operand lives where in the
// carve-out of one workgroup, for which problem dimensions.  Kept apart from the kernel so that the plan can be checked without a GPU
// (tests/test_ric_layout.py compiles it with g++ and checks that regions that are live together never overlap).
#pragma once
#ifndef RIC_MAGIC_DIV_DEFINED
#define RIC_MAGIC_DIV_DEFINED
// x / d for 0 <= x < 2^32 / d with mg = ceil(2^32 / d) (device_common.h has the device side)
static inline constexpr unsigned ric_magic_div(int d) { return (unsigned)((0x100000000ull + (unsigned long long)d - 1) / (unsigned long long)d); }
#endif

struct RicLds {
  int np, mp, nzp, ldl, ldr, nb, nbm, lw, nwb, gfull, st_lds, ovl, sq, nv;  // padded dims, leading dims of L / Lr, block counts, W leading dim / col blocks
  int PT, R1, LP, LI, AB, GP, vec, iwork, total_bytes;
  // KKT / value-update workspace carved from R1 once AB is dead
  int Lr, LIr, W, ST, CT, VX, Y, SC, LIs;
  unsigned mg_mp;  // magic_div(mp)
  int K2;  // legs: W2 (mp x (mp+1)) | VX2 (16 x (mp+1)) of the [I; 0] solve — inside the PT region when it fits (dead during step 6)
};

static inline constexpr RicLds make_ric_lds(int n, int m, int c, int gfull = 1, int st_lds = 1) {
  RicLds s{};
  s.st_lds = st_lds;  // 0: Sh^T (mp x np) lives in the L2-resident per-instance scratch instead of LDS (large m)
  s.gfull = gfull;  // 1: G = Pt [A B] kept whole (x part over PT, u part in GP) — needs np x mp doubles for GP instead of np x 16 ;
                    // 2: whole G with its u part in the L2-resident scratch (large m: only the few Ruu tiles read it back) ;
                    // 3: as 1 for a STRUCTURED problem (SQ kernels only): the sweep touches rows ks = (n / 2) & ~3 .. np of [A B] and nothing else, so the
                    //    region holds just those (S.AB points ks rows in front of it) — 36 of 80 rows freed at n = 76, which is what lets G_u and the
                    //    factor of Ruu stay on chip at m = 44 (round 5)
  const bool g1 = gfull == 1 || gfull == 3;
  const int ab_skip = gfull == 3 ? ((n / 2) & ~3) : 0;  // rows of [A B] in front of the first one the structured sweep reads
  s.np = (n + 15) & ~15; s.mp = (m + 15) & ~15; s.nzp = s.np + s.mp; s.ldl = s.np + 1; s.ldr = s.mp + 1;
  s.nb = s.np / 16; s.nbm = s.mp / 16;
// Odd leading dimensions for W / CT / VX (np + 17) and Y = Da^T (17) since round 5: the A-operand fetch of  W -= Y VX  walks the ROWS of Y with a
// stride of 16 doubles (sixteen lanes on one pair of banks), the row-wise passes over W / CT / VX likewise on 96.  SQ_LDS_BANK_CONFLICT /
// SQ_LDS_IDX_ACTIVE of the sweep 18.1 % -> 14.9 %, 1.607 -> 1.573 ms per launch (profiles/r05_lds_padding.txt); -DRIC_LW_PAD=0 -DRIC_LDY=16: as before.
#ifndef RIC_LW_PAD
#define RIC_LW_PAD 1
#endif
#ifndef RIC_LDY
#define RIC_LDY 17
#endif
  s.lw = s.np + 16 + RIC_LW_PAD;  // W = [K | pad | k | pad]: x-columns at 0..n-1, the feed-forward column at np
  s.nwb = (s.np + 16) / 16;
  int o = 0;
  auto take = [&](int cnt) { int r = o; o += (cnt + 1) & ~1; return r; };
  s.PT = take(s.np * (s.np + 1));  // leading dimension np + 1: conflict-free row AND column access
  s.R1 = o;
  s.LP = take(s.np * s.ldl); s.LI = take(s.nb * 272);             // phase 1 view of R1
  const int end1 = o;
  o = s.R1;
  s.AB = take((s.np - ab_skip) * s.nzp) - ab_skip * s.nzp; s.GP = take(g1 ? s.np * s.mp : (s.np * 16 > 8 * s.nzp ? s.np * 16 : 8 * s.nzp));               // phase 2 view (overlaps phase 1)
  const int end2 = o;
  o = s.R1;                                                         // phase 3 view (overlaps AB)
  s.Lr = take(s.mp * s.ldr); s.LIr = take(s.nbm * 272); s.W = take(s.mp * s.lw); s.ST = take(st_lds ? s.mp * s.np : 0);
  s.CT = take(16 * s.lw); s.VX = take(16 * s.lw); s.Y = take(s.mp * RIC_LDY); s.SC = take(16 * 17); s.LIs = take(272);
  int end3 = o;
  // Overlap layout: Lr / LIr in the G_u region (dead once the Ruu tiles are done) instead of on top of [A B] — the one-wavefront
  // factorisation of Ruu then runs while the other wavefronts still multiply [A B]^T G_x for the x rows of Hh.  The other
  // phase-3 operands are written after those tiles and may lie over [A B] and the head of G_u.
  s.ovl = 0;
  if (gfull == 3 && s.nbm <= 3) {
    // compressed [A B]: the phase-3 operands first (over [A B] and G_u: written once both are dead), Lr / LIr behind them — past the end of [A B], which the
    // other wavefronts still read while wavefront 0 factorises; over the tail of G_u at most, which is dead by then
    int cur = s.R1;
    auto place = [&](int cnt) { const int r = cur; cur += (cnt + 1) & ~1; return r; };
    const int w_ = place(s.mp * s.lw), st_ = place(st_lds ? s.mp * s.np : 0), ct_ = place(16 * s.lw), vx_ = place(16 * s.lw), y_ = place(s.mp * RIC_LDY),
              sc_ = place(16 * 17), lis_ = place(272);
    const int lr = cur > s.GP ? cur : s.GP, lir = lr + ((s.mp * s.ldr + 1) & ~1);  // (s.GP = the end of [A B])
    s.ovl = 1;
    s.Lr = lr; s.LIr = lir; s.W = w_; s.ST = st_; s.CT = ct_; s.VX = vx_; s.Y = y_; s.SC = sc_; s.LIs = lis_;
    end3 = lir + ((s.nbm * 272 + 1) & ~1);
  }
  if (gfull == 1 && s.nbm <= 3) {
    const int need = ((s.mp * s.ldr + 1) & ~1) + ((s.nbm * 272 + 1) & ~1);
    const int lr = s.GP + ((s.np * s.mp - need) & ~1), lir = lr + ((s.mp * s.ldr + 1) & ~1);  // at the end of the G_u region
    int cur = s.R1;
    auto place = [&](int cnt) { const int r = cur; cur += (cnt + 1) & ~1; return r; };
    const int w_ = place(s.mp * s.lw), st_ = place(st_lds ? s.mp * s.np : 0), ct_ = place(16 * s.lw), vx_ = place(16 * s.lw), y_ = place(s.mp * RIC_LDY),
              sc_ = place(16 * 17), lis_ = place(272);
    if (need <= s.np * s.mp && cur <= lr) {  // everything else fits in front of it: no growth of the carve-out
      s.ovl = 1;
      s.Lr = lr; s.LIr = lir; s.W = w_; s.ST = st_; s.CT = ct_; s.VX = vx_; s.Y = y_; s.SC = sc_; s.LIs = lis_;
      end3 = s.GP + s.np * s.mp;
    }
  }
  o = end1 > end2 ? end1 : end2;
  if (end3 > o) o = end3;
  s.K2 = s.PT;
  if (s.np * (s.np + 1) < (s.mp + 16) * (s.mp + 1)) s.K2 = take((s.mp + 16) * (s.mp + 1));  // small problems: own space
  s.vec = take(7 * s.nzp + 2 * c + 96 + 80);  // ph ft vv w gh pvec (6 nzp) | dtl kvc (2 c) | e6l wred t6l (92) | gpre (nzp) | d12l (74)
  s.mg_mp = ric_magic_div(s.mp);
  s.sq = 0; s.nv = 0;  // structured [A B] (set by the caller for whole-body problems, see step 5)
  s.iwork = o;
  s.total_bytes = o * 8 + (c + 72) * 4;
  return s;
}
