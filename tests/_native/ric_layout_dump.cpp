// Host-side dump of the sweep's LDS plans (mpc_benchmark_amd/csrc/riccati_layout.h) for tests/test_ric_layout.py: one line per plan,
// every region as name:start:length (doubles).
#include <cstdio>
#include "../../mpc_benchmark_amd/csrc/riccati_layout.h"

int main() {
  const int dims[][2] = {{76, 32}, {76, 44}, {56, 22}, {56, 34}, {76, 48}, {40, 20}, {24, 12}, {18, 12}};
  for (const auto& d : dims)
    for (int plan = 0; plan <= 3; ++plan)
      for (int st = 0; st <= 1; ++st) {
        const int n = d[0], m = d[1], c = 40;
        const RicLds s = make_ric_lds(n, m, c, plan, st);
        const int skip = plan == 3 ? ((n / 2) & ~3) : 0;
        const int gp = (plan == 1 || plan == 3) ? s.np * s.mp : (s.np * 16 > 8 * s.nzp ? s.np * 16 : 8 * s.nzp);
        printf("n=%d m=%d c=%d plan=%d st=%d total_bytes=%d ovl=%d iwork=%d np=%d mp=%d nzp=%d skip=%d | PT:%d:%d R1:%d:0 ABlive:%d:%d GP:%d:%d LP:%d:%d LI:%d:%d "
               "Lr:%d:%d LIr:%d:%d W:%d:%d ST:%d:%d CT:%d:%d VX:%d:%d Y:%d:%d SC:%d:%d LIs:%d:%d vec:%d:%d\n",
               n, m, c, plan, st, s.total_bytes, s.ovl, s.iwork, s.np, s.mp, s.nzp, skip, s.PT, s.np * (s.np + 1), s.R1, s.AB + skip * s.nzp, (s.np - skip) * s.nzp, s.GP, gp,
               s.LP, s.np * s.ldl, s.LI, s.nb * 272, s.Lr, s.mp * s.ldr, s.LIr, s.nbm * 272, s.W, s.mp * s.lw, s.ST, st ? s.mp * s.np : 0, s.CT, 16 * s.lw, s.VX, 16 * s.lw,
               s.Y, s.mp * RIC_LDY, s.SC, 16 * 17, s.LIs, 272, s.vec, 7 * s.nzp + 2 * c + 96 + 80);
      }
  return 0;
}
