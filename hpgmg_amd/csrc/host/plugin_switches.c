/*
 * plugin_switches.c -- the run-time switches of the operator plugin, in ONE table.
 *
 * The reference fixes its variants with -D flags at compile time; the plugin's own choices -- which fused form runs, whether
 * an exchange is overlapped, which debugging path is taken -- are run-time switches instead: each has an environment variable (read once, on
 * first use), a default, and a setter in the API (include/hpgmg_fv.h: hpgmg_set_*) that the tests use.  None of them changes a result; they
 * select between bit-identical ways of computing it.  HPGMG_SWITCHES=1 prints the table (with the values in force) when the library is loaded.
 */
#include "plugin_internal.h"

enum { K_ON = 0, K_OFF = 1, K_INT = 2, K_TRI = 3 };      /* on unless "0" / off unless "1" / an integer / "0" off, "1" on, anything else: the default (2) */
static struct { const char *env; int kind; long long dflt; const char *what; long long value; int known; } table[SW_COUNT] = {
  [SW_GHOST_FREE]        = { "HPGMG_GHOST_FREE", K_ON, 1, "stencil launches read neighbouring boxes and the Dirichlet rule themselves; 0: the reference's exchange + boundary + stencil launches" },
  [SW_ONE_LAUNCH_GHOSTS] = { "HPGMG_ONE_LAUNCH_GHOSTS", K_ON, 1, "27-pt / fv2 / fv4: box-to-box ghost copies and boundary conditions as one launch" },
  [SW_OVERLAP]           = { "HPGMG_OVERLAP", K_ON, 1, "N > 1: halo exchanges on a second stream under the part of the stencil launch that does not need them" },
  [SW_PAIR_REMOTE]       = { "HPGMG_PAIR_REMOTE", K_ON, 1, "N > 1, 7-pt: sweep pairs across rank boundaries (two-deep halo, one exchange per pair); 0: one exchange per sweep" },
  [SW_IMAGES]            = { "HPGMG_IMAGES", K_ON, 1, "N > 1, 27-pt / fv4: images of the neighbouring ranks' boxes (one-pass red + black kernels keep running); 0: exchange_boundary per half sweep" },
  [SW_FUSED_SWEEPS]      = { "HPGMG_FUSED_SWEEPS", K_ON, 1, "7-pt: two smoother sweeps per pass on bandwidth-bound levels" },
  [SW_PAIR_MIN_CELLS]    = { "HPGMG_PAIR_MIN_CELLS", K_INT, 2000000, "7-pt: smallest level (cells) that takes the sweep-pair kernel (2 M: the 128^3 level too, since the kernel picks its launch shape per launch -- 16 workgroup rows of 10 waves x 16 chunks = 256 workgroups there: config 2 2.70 vs 2.74 ms; with 16 waves it lost, 3.02 vs 2.92: profiles/r06k_ab_pair_min.txt, r06n_ab_pair_min.txt)" },
  [SW_FUSED_RESIDUAL]    = { "HPGMG_FUSED_RESIDUAL", K_ON, 1, "residual + restriction (+ zero_vector), residual + norm, norm + copy + restriction as one pass each" },
  [SW_FUSED_TAIL]        = { "HPGMG_FUSED_TAIL", K_ON, 1, "7-pt: the V-cycle below the brick levels (<= 8^3; <= 16^3 without them) as one single-workgroup launch" },
  [SW_FUSED_FTAIL]       = { "HPGMG_FUSED_FTAIL", K_ON, 1, "7-pt: the F-cycle's own work below 32^3 as one launch" },
  [SW_FUSED_BOTTOM]      = { "HPGMG_FUSED_BOTTOM", K_ON, 1, "BiCGStab bottom solve on the device; 0: driven from the host (host/solvers.c) through the operators" },
  [SW_SMALL_FUSED]       = { "HPGMG_SMALL_FUSED", K_TRI, 2, "27-pt / fv2 / fv4: smooth() of a one-box level as one launch on an LDS image (2, default); 0 off" },
  [SW_SMALL_VTAIL]       = { "HPGMG_SMALL_VTAIL", K_TRI, 2, "27-pt / fv2 / fv4: the V-cycle below a one-box level as one launch: 2 on except for 27-pt GSRB (default), 1 on, 0 off" },
  [SW_SMALL_27PT_GSRB]   = { "HPGMG_TUNE_SMALL_27PT_GSRB", K_OFF, 0, "27-pt GSRB: the one-launch small-level smooth() instead of the one-workgroup-per-box red + black kernel" },
  [SW_SMALL_OPS]         = { "HPGMG_SMALL_OPS", K_ON, 1, "BLAS-1 / operator calls of a host-driven Krylov solver on a small one-box level go out with the scalar that follows them" },
  [SW_LAZY]              = { "HPGMG_LAZY", K_ON, 1, "void operators are postponed while they follow MGVCycle's call order and issued fused (the reference's own driver, INTEGRATION.md Route B)" },
  [SW_LAZY_REPORT]       = { "HPGMG_LAZY_REPORT", K_OFF, 0, "print what the operator queue did when the process ends" },
  [SW_TEMP_SCRATCH]      = { "HPGMG_TEMP_SCRATCH", K_ON, 1, "inside a cycle VECTOR_TEMP is scratch after smooth(): in-cycle smoother forms; 0: the exact state of smooth() everywhere" },
  [SW_FV4_NO_EXACT_RB]   = { "HPGMG_TUNE_FV4_NO_EXACT_RB", K_OFF, 0, "fv4: the exported smooth() as six half sweeps (no red + black passes)" },
  [SW_GRAPH]             = { "HPGMG_GRAPH", K_OFF, 0, "capture / replay the launch-bound segments as hipGraphs (measured slower with a full stream)" },
  [SW_DEFER_NORM]        = { "HPGMG_DEFER_NORM", K_ON, 1, "FMGSolve: norm(F) of the opening pass is collected at the end, where it is used, instead of waited for at the start" },
  [SW_BRICK_VISITS]     = { "HPGMG_BRICK_VISITS", K_ON, 1, "7-pt: the 64^3 / 32^3 / 16^3 levels of a V-cycle leg as one launch of 8^3 bricks that trade faces, residuals and corrections inside the launch; 0 when other processes run such launches on the same GPU" },
  [SW_BRICK_SIZE]       = { "HPGMG_TUNE_BRICK", K_INT, 8, "side of those bricks: 8 (512 lanes, one cell each) or 16 (1024 lanes, four cells each)" },
  [SW_BRICK_MIN_DIM]    = { "HPGMG_TUNE_BRICK_MIN", K_INT, 16, "smallest level (cells per side) visited as bricks; the levels below it are the single-workgroup tail's" },
  [SW_BRICK_FSTEP]      = { "HPGMG_TUNE_BRICK_FSTEP", K_ON, 1, "FMGSolve: interpolation_fcycle onto a brick level rides in the first launch of the V-cycle that follows it" },
  [SW_BRICK_CHAIN]      = { "HPGMG_TUNE_BRICK_CHAIN", K_INT, 1, "consecutive brick levels of a V-cycle leg in ONE launch (what passes between the levels passes inside it); 0: one launch per level visit" },
  [SW_BRICK_WIDE]       = { "HPGMG_BRICK_WIDE", K_ON, 1, "27-pt / fv4: the same for the operators with wide stencils (kernels/brick_wide.hip: halo of the stencil's radius, apply_BCs_p2 / _v4 on the LDS image)" },
  [SW_BRICK_WIDE_MAX_DIM] = { "HPGMG_TUNE_BRICK_WIDE_MAX", K_INT, 64, "largest level (cells per side) the 27-pt / fv4 plugins visit as bricks (measured: fv4 `7 64` 29.71 / 29.91 / 30.15 ms with 64 / 32 / 16, 27-pt 12.27 / 12.51 / 12.52)" },
  [SW_BRICK_WIDE_TAIL_DIM] = { "HPGMG_TUNE_BRICK_WIDE_TAIL", K_INT, 1, "27-pt / fv4: the largest one-box level left to the single-workgroup tail (1: the 8^3, 4^3 (and 27-pt: 2^3) levels are visited as ONE brick each and only the bottom solve is left; 2 / 4 / 8: the tail starts there)" },
  [SW_FTAIL_MAX_DIM]    = { "HPGMG_TUNE_FTAIL_MAX", K_INT, 8, "7-pt: the largest level (cells per side) the single-launch F-cycle tail starts from (8: the 16^3 step of the climb is a brick launch each way, eight CUs instead of one: config 1 0.320 vs 0.328 ms with 16, config 2 the same within noise; profiles/r06i_ab_ftail.txt)" },
  [SW_SMOOTHER_PRECISION]= { "HPGMG_SMOOTHER_PRECISION", K_INT, 64, "32: fp32 coefficient streams in the Chebyshev sweep pairs (BASELINE config 5, tolerance-gated); 64: bit-exact" },
};

long long hp_switch(hp_switch_id id) {
  if (!table[id].known) {
    const char *e = getenv(table[id].env);
    long long v = table[id].dflt;
    switch (table[id].kind) {
      case K_ON:  v = !(e && e[0] == '0'); break;
      case K_OFF: v = (e && e[0] == '1'); break;
      case K_INT: if (e && *e) v = atoll(e); break;
      default:    v = (e && e[0] == '0') ? 0 : ((e && e[0] == '1') ? 1 : table[id].dflt); break;
    }
    table[id].value = v; table[id].known = 1;
  }
  return table[id].value;
}
void hp_switch_set(hp_switch_id id, long long value) { table[id].value = value; table[id].known = 1; }

void hpgmg_print_switches(void) {
  int q;
  for (q = 0; q < SW_COUNT; q++) fprintf(stderr, "  %-28s = %-8lld (default %lld)  %s\n", table[q].env, hp_switch((hp_switch_id)q), table[q].dflt, table[q].what);
}
__attribute__((constructor)) static void switches_report(void) { const char *e = getenv("HPGMG_SWITCHES"); if (e && e[0] == '1') hpgmg_print_switches(); }
