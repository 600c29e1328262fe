/*
 * include/flipv.h -- C-ABI of the MI355X-native FLIP substep (libflipv.so).
 *
 * The reference (rlguy/FLIPViscosity3D) has no FFI; its seam is the C++ class surface
 * FluidSimulation (reference fluidsimulation.h:53-63) and the operator parameter structs below
 * it.  This header is the drop-in boundary a host binds instead: plain pointers and sizes, no
 * C++/torch types, status codes instead of exceptions/abort().  Every entry point cites the
 * reference interface it replaces.
 *
 * Conventions
 *  - All grid pointers are HOST pointers in the reference's Array3d layout,
 *    flat = i + width*(j + height*k) (reference array3d.h:397-400), fp32, caller-owned.
 *    For an I x J x K grid:  U (I+1,J,K)  V (I,J+1,K)  W (I,J,K+1)  (macvelocityfield.cpp:40-48)
 *    liquid phi / pressure (I,J,K); solid phi / viscosity nodes (I+1,J+1,K+1).
 *  - Particles are AoS {px,py,pz,vx,vy,vz} fp32 = FluidParticle (fluidsimulation.h:39-48).
 *  - Return value: 0 = ok, > 0 = completed with a solver warning (not converged; result is still
 *    usable, like the reference which only prints), < 0 = error (flipv_last_error()).
 *    Nothing throws or aborts across this ABI.
 *  - One context per host thread; contexts are independent (one HIP stream each).
 *  - There is NO CPU fallback: without a HIP device flipv_create() fails with FLIPV_ERR_NO_DEVICE.
 *  - The library reads NO environment variables: every switch that changes what a solve does is a field of flipv_params.
 */
#ifndef FLIPV_H
#define FLIPV_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FLIPV_VERSION 6   /* 3: every behavioural switch is a flipv_params field (no environment variables); brick layout; residual replacement
                             4: the constants of the two-stage viscosity solve are flipv_params fields; flipv_solve_info reports the correction stage
                             5: the velocity criterion and the mass scale of the viscosity solve (flipv_solve_info.velocity_step); A/B and measurement switches moved to
                                flipv_debug_params; residual_replacement removed; flipv_abi_version() */

typedef struct flipv_context flipv_context;

enum flipv_status {
    FLIPV_OK = 0,
    FLIPV_WARN_NOT_CONVERGED = 1,   /* cap reached, result applied (viscositysolver.cpp:680-683, pressuresolver.cpp:564-566) */
    FLIPV_WARN_SOLVE_FAILED = 2,    /* viscosity rejected, velocity left untouched (fluidsimulation.cpp:195) */
    FLIPV_ERR_INVALID = -1,
    FLIPV_ERR_NO_DEVICE = -2,
    FLIPV_ERR_HIP = -3,
    FLIPV_ERR_OOM = -4,
    FLIPV_ERR_COMM = -5
};

/* Grid ids for flipv_read_grid / flipv_write_grid.  Valid masks travel as 0/1 floats. */
enum flipv_grid {
    FLIPV_GRID_U = 0, FLIPV_GRID_V = 1, FLIPV_GRID_W = 2,                   /* _MACVelocity        fluidsimulation.h:116 */
    FLIPV_GRID_SAVED_U = 3, FLIPV_GRID_SAVED_V = 4, FLIPV_GRID_SAVED_W = 5, /* _savedVelocityField fluidsimulation.h:117 */
    FLIPV_GRID_VALID_U = 6, FLIPV_GRID_VALID_V = 7, FLIPV_GRID_VALID_W = 8, /* _validVelocities    fluidsimulation.h:118 */
    FLIPV_GRID_LIQUID_PHI = 9,                                              /* _liquidSDF          fluidsimulation.h:123 */
    FLIPV_GRID_SOLID_PHI = 10,                                              /* _solidSDF           fluidsimulation.h:120 */
    FLIPV_GRID_WEIGHT_U = 11, FLIPV_GRID_WEIGHT_V = 12, FLIPV_GRID_WEIGHT_W = 13, /* _weightGrid   fluidsimulation.h:126 */
    FLIPV_GRID_VISCOSITY = 14,                                              /* _viscosity          fluidsimulation.h:132 */
    FLIPV_GRID_PRESSURE = 15,                                               /* PressureSolver::solve result, pressuresolver.h:177 */
    FLIPV_GRID_COUNT = 16
};

enum flipv_preconditioner { FLIPV_PRECOND_AUTO = 0, FLIPV_PRECOND_DIAGONAL = 1, FLIPV_PRECOND_MULTIGRID = 2 };

enum flipv_precision {
    FLIPV_PRECISION_FP32 = 0, /* solver vectors fp32, every reduction and scalar fp64 (default) */
    FLIPV_PRECISION_FP64 = 1  /* solver vectors fp64 like the reference's VectorXd / std::vector<double>.  Pressure and the diagonally preconditioned viscosity solve: fp64
                                 PCG vectors.  Viscosity under the multigrid (an fp32 V-cycle): mixed-precision iterative refinement -- solution and residual b - A_ref x in
                                 fp64, the Krylov loops in fp32, repeated until the FP64 residual meets viscosity_tolerance (status 1 if it does not) */
};

/* THE DEFAULT VISCOSITY SOLVE, stated once (k_viscosity.hip: viscosity_solve_t implements this table; DESIGN.md 4 gives the measurements behind every number).
 *   S = nu_max dt/dx^2, the a-priori stiffness (all ranks).                     N = the norm every tolerance is a share of = min(max|rhs|, max(viscosity_mass_scale x max|u|, viscosity_mass_floor x max|rhs|))
 *   before the solve         rows the reference's system determines only through its solver's start at 0 are taken OUT of the system and held at 0 (flipv_solve_info.eliminated_rows):
 *                            (i) of two or more rows without own volume that are THE SAME equation (their only non-zero factor is one edge's / cell centre's) all but the first in the
 *                            reference's row order -- its MIC(0) factorisation zeroes the later ones' pivots and couplings (pcgsolver.h:62-178) --; (ii) connected sets of such rows that
 *                            share no stress term with a row that has own volume or with a solid face (zero right-hand side, singular in the exact operator)   [FLIPV_VERSION 6]
 *   preconditioner (AUTO)    S <= 8: diagonal PCG on the reference's operator A_ref (one loop to viscosity_tolerance x N).   S > 8: Galerkin multigrid V(2,2), unless
 *                            the previous solve shows the diagonal to converge for less; a diagonal solve AUTO picked that hits the cap is repeated with the multigrid
 *   under the multigrid      defect correction towards A_ref = A + E (E: the rounding of the reference's float diagonal; with a viscosity FIELD also its per-row edge factors):
 *     stage 1                PCG on the exact operator A, right-hand side b - E u_old (viscosity_defect_predictor), to viscosity_stage1_factor x viscosity_tolerance x N:
 *                            factor 300 for S <= 1 000, 3 000 for 1 000 < S <= viscosity_two_stage_max_stiffness (1e6; 2e5 until round 5: at 256^3 / nu = 500,
 *                            S = 3.3e5, the early stop halves the iterations and ends the capped solves), 1 beyond (and no predictor there)
 *     correction stage(s)    x flushed to fp64, r = b - A_ref x in fp64, PCG on A dx = r to max(floor, share x (|r| - stage 1's target)), share = viscosity_stage2_factor:
 *                            1e-2 for S <= 2e4, 1e-3 beyond (2e-2 behind a stage 1 that ran to the tolerance); floor = max(viscosity_tolerance x N, 1e-3 x stage 1's target);
 *                            at most viscosity_stage2_max_iterations (200) each; viscosity_stage2_rounds = 1 stage -- 2 where the viscosity field's contrast max / min exceeds 1e4
 *                            (zero included); a stage that RAISES the fp64 residual is taken back, one that ends short of its target is restarted once
 *   the delivering loop      (a correction stage, its restart, or the one loop of a solve without stages) also needs the VELOCITY CRITERION: its last
 *                            viscosity_velocity_window (4) iterations together moved no velocity the substep uses by more than viscosity_velocity_tolerance (3e-5; 1e-5 where S > 5e4) x max|u|;
 *                            (held for at most 48 iterations past the residual test; optional early way out: viscosity_velocity_stall_ratio, off by default);
 *                            a loop the stall guard stops with the criterion unmet is restarted from the fp64 residual
 *   a viscosity FIELD        (values that differ) also gets: the weak modes of strongly coupled PAIRS of rows in the preconditioner (viscosity_pair_correction: coupling >= 0.7 of the geometric
 *                            mean of the diagonals, the pair's other couplings no larger than its weak eigenvalue; each adds what the V(2,2) smoother leaves of v v^T / lambda), and a stall
 *                            guard of 1 000 x instead of 16 x (CG's max|r| rebounds 20-50 x while it resolves the isolated modes of a viscosity jump)   [FLIPV_VERSION 6]
 *   after the solve          clusters of <= 4 rows without own volume that share one dominant stress term are solved exactly in fp64 (viscosity_massless_polish)
 *   status                   0 = every stage reached its target; 1 = cap / stalled / a stage ended short or was taken back (the result is applied, like the reference's
 *                            accepted iterate); flipv_solve_info: residual (stage 1's), defect_residual = max|b - A_ref x| delivered, velocity_step, correction_* */

/* Tunables.  Defaults = the reference's private constants (fluidsimulation.h:121,128-130,
 * pressuresolver.h:224-226, viscositysolver.h:200-202).  Every field below check_every: 0 = the default behaviour (a zero-initialised tail is a valid
 * default configuration).  Switches that exist for A/B measurements and tests live in flipv_debug_params, not here. */
typedef struct flipv_params {
    float cfl_number;            /* 5.0   _CFLConditionNumber */
    float min_frac;              /* 0.01  _minfrac */
    float pic_ratio;             /* 0.05  _ratioPICtoFLIP */
    int extrapolation_layers;    /* 0 => ceil(cfl_number)+2 (fluidsimulation.cpp:692) */
    double pressure_tolerance;   /* 1e-9 absolute inf-norm of the residual (pressuresolver.cpp:544) */
    double pressure_rel_tolerance; /* additional floor: tol = max(pressure_tolerance, rel * max|b|); fp32 vectors cannot
                                      reach 1e-9 absolute when |b| ~ 1; default 1e-6 for FP32, 0 for FP64 */
    int pressure_max_iterations; /* 200 in the reference (MIC(0)); the GPU preconditioner needs more iterations for the
                                    same residual, default 2000 */
    double viscosity_tolerance;  /* 1e-6 relative to the norm N of the table above (pcgsolver.h:259: max|rhs|) */
    int viscosity_max_iterations;/* 700 (viscositysolver.h:202) */
    double viscosity_accept_tolerance; /* 10.0 (viscositysolver.h:201) */
    int precision;               /* enum flipv_precision */
    int check_every;             /* convergence poll interval in iterations; 0 (default) = 32 on one GPU, 8 with a communicator (the diagonal loops), 4 in the
                                    multigrid-preconditioned loops (viscosity and pressure: an iteration after the stop is a full V-cycle) */
    int pressure_preconditioner; /* enum flipv_preconditioner; AUTO = aggregation multigrid with fp32 vectors on grids above 16^3, the diagonal otherwise */
    int viscosity_preconditioner;/* enum flipv_preconditioner; the table above.  Decisions use iteration counts and all-reduced scalars only, never timings: every rank
                                    of a communicator decides alike, and a run is reproducible up to summation order */
    int exact_viscosity_operator;/* 0 (default): the solve is for the REFERENCE's operator A_ref, including the rounding of its float diagonal (the reference sums
                                    vol + fR + fL + fT + fB + fF + fK in float, viscositysolver.cpp:394-446) -- at 256^3 (nu dt/dx^2 = 3 300) the reference's converged
                                    velocities are 7e-6 from this operator's and 1.5e-4 from the exact one's.  1: the exact operator vol u - div(tau) (better
                                    conditioned; what rounds differently is the reference): one multigrid-PCG loop, no defect correction.
                                    REPRODUCIBILITY: scatters and dot products sum in arrival order, so two runs agree to solver tolerance, not bit for bit.  With the
                                    reference's operator that can be coarser LOCALLY: its rounded diagonal leaves the near-rigid modes of tiny detached liquid clusters
                                    (own volumes of the size of the defect, which may even come out slightly negative) ill-determined -- two runs of one configuration
                                    were seen 4e-2 apart on the faces of such a cluster on the substep where a body touches the wall; the reference itself has the same
                                    indeterminacy, it only sums in a fixed order.  exact_viscosity_operator = 1 does not */
    int viscosity_layout;        /* layout of the viscosity solver's arrays: 0 = chosen per solve (bricks of 8 x 4 x 2 indices on sparse
                                    liquids -- rows filling < 30-40 % of the box, over all ranks --, plain planes otherwise), 1 = plain planes,
                                    2 = plain planes with the own-index arrays in 8 x 4 patches under the 16-lane tile geometry, 3 = bricks always */
    int multigrid_rank_local;    /* several ranks (block contexts).  0 (default): both multigrid preconditioners are the single domain's -- the fine-level sweeps read
                                    the neighbours' current values, the coarse hierarchy is the GLOBAL one: the iteration counts of a single domain.
                                    1: every rank cycles a hierarchy of its OWN rows / cells with the couplings across the cuts dropped and no exchange
                                    (block-Jacobi): 3-4x the viscosity iterations and 2-3x the pressure iterations on 2x2x2 blocks */
    int multigrid_distributed_levels; /* several ranks, viscosity multigrid with the global hierarchy: 0 = level 1 is DISTRIBUTED (cycled by the rows' owners with
                                    1-entry halo exchanges, like level 0) where the system has more than 4.5e6 rows over all ranks; 1 = always; -1 = never */
    int verbose;                 /* 1: one line per viscosity solve and the multigrid's level table on stderr; 2: also the residual history */
    /* the two-stage viscosity solve (the table above); 0 = the default in brackets */
    float viscosity_stage1_factor;          /* [300 | 3 000] >= 1; 1 = the strict solve: stage 1 to viscosity_tolerance itself (bench.py: mode_b_strict) */
    float viscosity_stage2_factor;          /* [1e-2 | 1e-3 | 2e-2] a correction stage's target as a share of the defect it starts from; 0 < share <= 0.5 */
    int viscosity_stage2_max_iterations;    /* [200] iteration budget of ONE correction stage (inside viscosity_max_iterations overall) */
    int viscosity_stage2_rounds;            /* [1 | 2] correction stages at most; 2 brings the velocities to <= 6e-6 of the reference's at every stiffness measured, for ~35 %
                                               more iterations than one stage at the default share */
    float viscosity_two_stage_max_stiffness;/* [1e6] nu dt/dx^2 up to which stage 1 stops early */
    int viscosity_defect_predictor;         /* [0 = on] -1 = off */
    /* what certifies the stop in late states (FLIPV_VERSION 5) */
    float viscosity_velocity_tolerance;     /* [0 = 3e-5, and 1e-5 where S > 5e4; -1 = off] the velocity criterion of the delivering loop.  The reference's test, max|r| <= 1e-6 max|rhs| (pcgsolver.h:259-272),
                                               does not bound the velocity error where the liquid holds light, weakly attached parts -- films and specks whose control volumes
                                               sum to a few per cent of a cell: residual = mass x error -- and in such states (the fringe of a splash, a body resting on the wall)
                                               CG still moves velocities by 1e-4 of their maximum per iteration when the residual test passes: the reference's own 1e-6 iterate is
                                               then 1e-4 ... 3e-1 of max|u| from the solution of its system on one substep in eight (profiles/r5/late_states.log).  Costs nothing
                                               where the iteration has settled when the residual passes (a compact falling body), 5-50 iterations in the states above */
    int viscosity_velocity_window;          /* [4] iterations the criterion sums over (1..8) */
    float viscosity_mass_scale;             /* [100; -1 = off] N = min(max|rhs|, viscosity_mass_scale x max|u|).  Once the liquid touches a wall max|rhs| is set by the rows next to
                                               solid faces (nu dt/dx^2 x the solid faces' velocities: 2 900 at 256^3 where max|u| = 1.4) and the reference's test no longer says
                                               anything about the bulk, whose near-rigid motions have residual = volume x error: round 4's rule left 3e-4 ... 9e-4 of max|u| on
                                               25 000 - 80 000 faces of the 256^3 bunny from the impact on (profiles/r5/eta_scan_256.log).  100 x 1e-6 = the final residual never
                                               above 1e-4 of a full control volume moving at max|u| */
    float viscosity_mass_floor;             /* [0 = by stiffness: max(0.03, min(0.3, 1e-5 S, 1e4 / S))] ... and N is never below this share of max|rhs|: a liquid almost at rest next to solid faces
                                               that still hold old velocities has max|u| / max|rhs| ~ 1e-5, and what an fp32 correction stage reaches scales with S: 256^3 honey
                                               (S = 32 768) settling on the floor ran 26 ... 86 of 330 solves out of their stage budget at 0.03, 2 at 0.3 (profiles/r5/mass_floor_scan.log) */
    int viscosity_massless_polish;          /* [0 = on; -1 = off] after the solve, clusters of <= 4 rows WITHOUT own volume that share one dominant stress term (>= 0.99 of each row's
                                               diagonal) are solved exactly in fp64 with everything around them held: their common stress is what the system determines, their split
                                               hangs on couplings 1e-5 of it -- a mode fp32 CG neither sees nor moves (holdout draws 20, 30: one used face 2e-4 ... 4e-4 from the
                                               reference's converged answer, carried to 8 ... 180 faces by the projection and the extrapolation; profiles/r5/holdout_misses_20_30.log) */
    float viscosity_velocity_stall_ratio;   /* [0 = off] OPT-IN early way out of the velocity criterion: the residual has passed, the last window moved the velocities by no more than
                                               10 x viscosity_velocity_tolerance, and by no less than this ratio (e.g. 0.5) x what the window before it moved.  On the 256^3 bunny lying
                                               on the wall 40 % of the solves sit on such a plateau -- 2e-5 ... 4e-4 max|u| per iteration on rows the system barely determines, while
                                               max|r| falls three more orders (profiles/r5/step_history_256.log) -- and run out the criterion's patience of 48 iterations: with 0.5 the
                                               first 400 substeps of that scene take 14.6 instead of 15.9 ms.  NOT the default: a plateau can also be a light part CG has not resolved YET
                                               (64^3, nu = 0.5, one substep of the impact: 9.5e-4 from the converged reference with the exit, 1e-6 without) and nothing in the
                                               iteration's history tells the two apart */
    int viscosity_pair_correction;          /* [0 = on where the viscosity is a FIELD] 1: on everywhere; -1: off.  Multigrid loops: pairs of rows whose coupling is >= 0.7 of the geometric mean of their diagonals -- a row (almost) without own
                                               volume hanging on one stress term and the row that shares it -- get their 2 x 2 block solved in every preconditioner application, added to the
                                               V-cycle (FLIPV_VERSION 6).  Each such pair carries a mode of Jacobi-scaled eigenvalue 1 - |coupling| (2e-5 ... 1e-3 on a viscosity field with a
                                               jump) that the geometric hierarchy does not see and CG otherwise resolves one plateau at a time */
} flipv_params;

/* Switches for A/B measurements, profiling and tests (flipv_set_debug_params).  Results do not depend on them beyond solver tolerance; none of them is needed to
 * run a simulation.  All zero = the product's behaviour. */
typedef struct flipv_debug_params {
    int kernel_timing;           /* 1 => bracket every SpMV launch with HIP events (flipv_kernel_stats) */
    int tile_rows;               /* 16 | 64: pins the tile geometry of the plane-layout kernels (lanes of a wave along i); 0 = chosen per solve */
    int viscosity_mg_coarsest_sweeps; /* sweeps on the LDS-resident coarsest level of the viscosity multigrid: a power of two (4..64) = that many Chebyshev-weighted Jacobi
                                    sweeps, any other count <= 64 = plain damped Jacobi sweeps; 0 = chosen per solve (8 under the two-stage solve) */
    int viscosity_mg_min_dim;    /* the viscosity hierarchy stops at the level whose longest axis is <= this many cells; 0 = 16 */
    int pressure_mg_coarsest_sweeps; /* 0 = 8; <= 64 */
    float pressure_mg_omega;     /* damping of the pressure multigrid's Jacobi sweeps; 0 = 0.9 */
    float pressure_mg_overcorrection; /* scaling of its coarse-grid correction; 0 = 1.8 */
    float viscosity_mg_omega_first;  /* damping of the first and of the second Jacobi sweep of the viscosity multigrid's V(2,2) smoother; 0 = the defaults */
    float viscosity_mg_omega_second;
    int no_liquid_box;           /* 1: every sweep of a substep covers the whole box instead of the neighbourhood of the liquid */
    int no_comm_overlap;         /* 1: the halo exchange of the PCG search direction does not overlap the interior SpMV */
    int no_graph_replay;         /* 1: the PCG loops are launched kernel by kernel instead of replayed as hipGraphs */
    int unbinned_scatter;        /* 1: particle scatters with global atomics instead of LDS tiles */
    int grid_cap;                /* n>0: cap of the PCG kernels' grids in blocks (tests: every block walks many tiles) */
    int viscosity_lane_width;    /* 2|4: forced lane width of the viscosity tile kernels (2 excludes the brick layout and the multigrid) */
    int viscosity_spmv_grid_cap; /* n>0: grid cap of the viscosity SpMV kernel alone */
    int viscosity_update_grid_cap; /* n>0: grid cap of the viscosity init/update kernels */
    int beta_from_conjugacy;     /* 1: the diagonal PCG's (r/d,q) is replaced by (s,q) -- equal in exact arithmetic -- and the SpMV skips the residual: 5-7 % faster per
                                    iteration, but on ill-conditioned systems the fp32 solve stagnates */
    int spmv_run_length;         /* k-marching SpMV kernels walk runs of up to this many tiles along k (2..64); 0 = chosen per solve; -1 = the tile-at-a-time kernels; -2 = the pressure
                                    SpMV's address-order sweep kernel (filled boxes in 64-lane rows) whatever the size */
    int viscosity_mg_packed_rows; /* coarse rows of the viscosity multigrid as the cycle reads them: 0 = chosen per solve (packed fp16 up to nu dt/dx^2 = 2e5), 1 = packed fp16, -1 = the fp32 grids */
    float stall_guard_ratio;     /* the stall guard of the PCG loops stops a loop whose max|r| exceeds this x the smallest it has reached (once that is within 100 x the tolerance); 0 = 16, and 1 000 in the viscosity solve of a viscosity FIELD (FLIPV_VERSION 6) */
    float viscosity_pair_lambda_floor; /* the pair correction's gain is 1 / max(lambda, this) per unit diagonal; 0 = 1e-5 */
    int velocity_patience;       /* iterations the velocity criterion holds a loop whose residual has passed before it lets it end (flipv_solve_info.velocity_step tells what was left); 0 = 48 */
} flipv_debug_params;

typedef struct flipv_solve_info {
    int iterations;      /* iterations run (count) */
    double residual;     /* final inf-norm residual */
    double rhs_norm;     /* max|rhs| */
    int status;          /* 0 converged (two-stage viscosity solve: every stage reached its target), 1 cap reached / a stage incomplete (the result is applied),
                            2 failed/rejected, 3 trivial (rhs ~ 0 or skipped) */
    int rows;            /* unknowns */
    int active_tiles;    /* tiles swept per launch */
    int total_tiles;
    int preconditioner;  /* 0 diagonal, 1 multigrid (pressure: aggregation V-cycle with fp32 vectors, rank-local under blocks; viscosity:
                            the Galerkin V-cycle) */
    int layout;          /* viscosity: layout of the solver's arrays in this solve, 0 plain planes, 1 plain + swizzled own-index arrays, 2 bricks
                            (active_tiles / total_tiles then count bricks of 8 x 4 x 2 indices) */
    int refinements;     /* viscosity, fp32 vectors: fp64 residual evaluations (x flushed into an fp64 accumulator, r = b - A x in fp64) -- after a stall
                            (the PCG is then restarted on the correction equation) and around every defect-correction stage; `iterations` counts all rounds */
    double defect_residual; /* viscosity, default operator under the multigrid: max|b - A_ref x| (fp64) that the solve DELIVERS, A_ref the
                            reference's float-rounded operator; `residual` is that of the exact-operator PCG loop (stage 1: it stops at
                            viscosity_stage1_factor x viscosity_tolerance where a correction stage follows, see exact_viscosity_operator).  0 when there
                            was no such stage */
    int correction_iterations; /* iterations spent in defect-correction stages (part of `iterations`) */
    double comm_bytes_setup;   /* several ranks, multigrid preconditioner with the global coarse hierarchy: bytes this solve ALL-REDUCED once (the first coarse level's
                                  operator) ... */
    double comm_bytes_per_iteration; /* ... and per iteration (that level's right-hand side); 0 on one rank / with the diagonal */
    double velocity_step;      /* viscosity: what the last viscosity_velocity_window iterations of the delivering loop moved, as a share of max|u| (the quantity the velocity
                                  criterion tests; 0 when the criterion is off) */
    int halo_exchanges_per_iteration; /* several ranks: neighbour exchanges (a halo copy or reduction to all <= 26 neighbours = 1) ... */
    int allreduces_per_iteration;     /* ... and all-reduces ONE iteration of this solve's (last) loop issued; 0 on one rank */
    int correction_status;     /* 0 no correction stage; 1 the (last) stage reached its target; 2 it ran into its iteration budget or stalled first -- also after the
                                  one restart such a stage gets -- (its result is kept if it lowered the fp64 residual; `status` is then 1); 3 the last stage RAISED the fp64 residual and was taken back
                                  (`status` 1) */
    int eliminated_rows;       /* viscosity: rows of the reference's system that REPEAT another row's equation (massless faces around one edge or cell centre whose only non-zero factor is that
                                  edge's: the reference's matrix is singular there) and were held at 0 instead of solved for -- what the reference's MIC(0)-PCG leaves them at (k_viscosity.hip:
                                  k_visc_singular_find).  `rows` + this = the reference's row count (FLIPV_VERSION 6) */
    int massless_cluster_edges; /* viscosity: control-volume edges with a massless cluster around them that were listed before the solve (viscosity_massless_polish); the list holds
                                  65 536: beyond that the rest are neither solved apart nor taken out of the velocity criterion's sight (FLIPV_VERSION 6) */
} flipv_solve_info;

/* Per-substep report (replaces the reference's stdout banners, fluidsimulation.cpp:143-163). */
enum { FLIPV_PHASE_SDF = 0, FLIPV_PHASE_P2G = 1, FLIPV_PHASE_BODYFORCE = 2, FLIPV_PHASE_VISCOSITY = 3,
       FLIPV_PHASE_PROJECT = 4, FLIPV_PHASE_CONSTRAIN = 5, FLIPV_PHASE_ADVECT = 6, FLIPV_PHASE_COUNT = 7 };

typedef struct flipv_stats {
    double phase_ms[FLIPV_PHASE_COUNT]; /* GPU time per phase (HIP events on the context stream) */
    double total_ms;
    float dt;                           /* substep size taken */
    int substeps;                       /* flipv_advance only */
    flipv_solve_info viscosity;
    flipv_solve_info pressure;
} flipv_stats;

/* Accumulated HIP-event timings of the two SpMV kernels since the last reset (kernel_timing=1). */
typedef struct flipv_kernel_stats {
    double pressure_spmv_ms;   long pressure_spmv_launches;   double pressure_spmv_cells;  /* cells swept, summed over launches */
    double viscosity_spmv_ms;  long viscosity_spmv_launches;  double viscosity_spmv_cells;
} flipv_kernel_stats;

/* ---- lifetime: FluidSimulation::initialize (fluidsimulation.cpp:26-43), without the boundary mesh ---- */
int flipv_create(int isize, int jsize, int ksize, float dx, flipv_context **out);
int flipv_create_on_device(int isize, int jsize, int ksize, float dx, int hip_device, flipv_context **out);
/* One rank of a block decomposition (no reference counterpart: the reference is single-process).
 * The context owns and computes the cells [cell_lo, cell_hi) of the GLOBAL I x J x K grid and ALLOCATES only that box plus a
 * halo of 8 entries on every side that has a neighbour: memory per rank scales with the block, not with the domain.
 * Blocks must form a tensor-product decomposition (the cuts along each axis are the same for every rank) and cuts along i
 * must be multiples of 8.  Particles uploaded to the context must lie in its cells.  A communicator must be attached
 * (flipv_comm_init_*) before the first substep when there is more than one rank.
 * Grids: flipv_read_grid / flipv_write_grid keep their full-size Array3d signature on such a context -- a read fills the
 * entries the rank owns and leaves the rest of the caller's array untouched, a write takes the entries the rank allocates
 * (owned + halo; the solid SDF and the viscosity have no exchange of their own, so their halo comes from the caller) -- and
 * flipv_read_grid_box / flipv_write_grid_box move the same data box-shaped, so that no full-size array has to exist on
 * the host either (flipv_grid_box gives the box: kind 0 = owned, what a read returns; kind 1 = allocated, what a write
 * takes; both clipped to the lattice of that grid, x fastest).
 * flipv_create_slab = a block that spans i and j: slabs along k. */
int flipv_create_block(int isize, int jsize, int ksize, float dx, int hip_device, const int *cell_lo, const int *cell_hi, flipv_context **out);
int flipv_block_range(flipv_context *ctx, int *cell_lo, int *cell_hi);
int flipv_create_slab(int isize, int jsize, int ksize, float dx, int hip_device, int k_begin, int k_end, flipv_context **out);
int flipv_slab_range(flipv_context *ctx, int *k_begin, int *k_end);
/* A single-domain context that serves ONLY the scene-setup entry points below (flipv_reset_boundary, flipv_add_boundary_mesh,
 * flipv_add_liquid_mesh, flipv_mesh_level_set) and the grid / particle transfers: 3 grids instead of ~70, i.e. ~4 % of the
 * memory of a full context.  This is how a rank of a block decomposition builds its scene on its own device: set the whole
 * scene up here (deterministic: every rank gets the same solid SDF and particles), hand the solid SDF to the block context
 * (flipv_write_grid takes the entries of its box) together with the particles that lie in its cells, destroy this one.
 * Every substep entry point returns FLIPV_ERR_INVALID on such a context. */
int flipv_create_setup(int isize, int jsize, int ksize, float dx, int hip_device, flipv_context **out);
int flipv_destroy(flipv_context *ctx);
const char *flipv_last_error(flipv_context *ctx); /* ctx may be NULL for create-time errors */
int flipv_device_name(flipv_context *ctx, char *buf, size_t len);

/* FLIPV_VERSION the library was built from: a binding compiled against another version of this header must not pass its structs (their layout changes with the
 * version: 4 -> 5 split flipv_params, added fields to flipv_solve_info) */
int flipv_abi_version(void);
int flipv_default_params(flipv_params *p);
int flipv_set_params(flipv_context *ctx, const flipv_params *p);   /* FLIPV_ERR_INVALID (flipv_last_error names the field) when a field is out of its documented range */
int flipv_get_params(flipv_context *ctx, flipv_params *p);
int flipv_default_debug_params(flipv_debug_params *p);
int flipv_set_debug_params(flipv_context *ctx, const flipv_debug_params *p);
int flipv_get_debug_params(flipv_context *ctx, flipv_debug_params *p);

/* setGravity (fluidsimulation.cpp:126-132) */
int flipv_set_gravity(flipv_context *ctx, float gx, float gy, float gz);
/* _solidSDF contents after addBoundary/resetBoundary (fluidsimulation.cpp:45-62): (I+1)(J+1)(K+1) nodes */
int flipv_set_solid_sdf(flipv_context *ctx, const float *nodes);
/* setViscosity(float) / setViscosity(Array3d<float>&) (fluidsimulation.cpp:99-124) */
int flipv_set_viscosity_uniform(flipv_context *ctx, float value);
int flipv_set_viscosity(flipv_context *ctx, const float *nodes);

/* the public `particles` vector (fluidsimulation.h:63) */
int flipv_upload_particles(flipv_context *ctx, const float *aos6, size_t n);
int flipv_download_particles(flipv_context *ctx, float *aos6, size_t capacity, size_t *n_out);
size_t flipv_num_particles(flipv_context *ctx);

size_t flipv_grid_elements(flipv_context *ctx, int which);
int flipv_read_grid(flipv_context *ctx, int which, float *out);
int flipv_write_grid(flipv_context *ctx, int which, const float *in);
int flipv_grid_box(flipv_context *ctx, int which, int kind, int *lo, int *hi);   /* kind 0 owned, 1 allocated; lo/hi: 3 ints each */
int flipv_read_grid_box(flipv_context *ctx, int which, float *out);              /* the owned box */
int flipv_write_grid_box(flipv_context *ctx, int which, const float *in);        /* the allocated box */
/* any box [lo, hi) of global indices inside what the context allocates of that grid, box-shaped (x fastest): how a rank takes ITS part of a scene
 * built on a setup context (flipv_create_setup) -- lo/hi from the block context's flipv_grid_box(kind 1) -- without a full-size host array */
int flipv_read_grid_region(flipv_context *ctx, int which, const int *lo, const int *hi, float *out);

/* ---- per-operator entry points, one per seam of advance() (fluidsimulation.cpp:138-167) ---- */
int flipv_cfl(flipv_context *ctx, float *dt_out);                 /* _cfl                      fluidsimulation.cpp:241-269 */
int flipv_particle_sdf(flipv_context *ctx);                       /* ParticleLevelSet::calculateSignedDistanceField particlelevelset.cpp:77-86 */
int flipv_p2g(flipv_context *ctx);                                /* _advectVelocityFieldU/V/W fluidsimulation.cpp:440-498 */
int flipv_extrapolate(flipv_context *ctx);                        /* extrapolateVelocityField  macvelocityfield.cpp:689-694 */
int flipv_save_velocity(flipv_context *ctx);                      /* _savedVelocityField = _MACVelocity  fluidsimulation.cpp:518 */
int flipv_advect_velocity_field(flipv_context *ctx);              /* _advectVelocityField      fluidsimulation.cpp:500-519 */
int flipv_body_force(flipv_context *ctx, float dt);               /* _addBodyForce             fluidsimulation.cpp:271-312 */
int flipv_viscosity_solve(flipv_context *ctx, float dt, flipv_solve_info *info); /* ViscositySolver::applyViscosityToVelocityField viscositysolver.cpp:41-63 */
int flipv_compute_weights(flipv_context *ctx);                    /* _computeWeights           fluidsimulation.cpp:549-582 */
int flipv_pressure_solve(flipv_context *ctx, float dt, flipv_solve_info *info);  /* PressureSolver::solve  pressuresolver.cpp:166-194 */
int flipv_apply_pressure(flipv_context *ctx, float dt);           /* _applyPressure            fluidsimulation.cpp:598-688 */
int flipv_constrain(flipv_context *ctx);                          /* _constrainVelocityField   fluidsimulation.cpp:696-729 */
/* NOTE: flipv_advect_particles is the fused particle kernel of the substep -- it performs the PIC/FLIP velocity update
 * (fluidsimulation.cpp:341-352) AND the RK2 advection (:315-339) in one pass over the particles, the order advance()
 * calls them in.  flipv_update_particle_velocities exists for parity tests of the velocity update alone; calling both
 * in sequence applies the velocity update twice. */
int flipv_update_particle_velocities(flipv_context *ctx);         /* _updateFluidParticleVelocities fluidsimulation.cpp:341-352 (alone) */
int flipv_advect_particles(flipv_context *ctx, float dt);         /* _updateFluidParticleVelocities + _advectFluidParticles  fluidsimulation.cpp:341-352, 315-339 */

/* ---- scene setup on the device (single-domain contexts) ----------------------------
 * Meshes are passed as in TriangleMesh (trianglemesh.h:36-37): `vertices` = nvertices x {x,y,z} floats, `triangles` =
 * ntriangles x 3 vertex indices; closed, consistently wound surfaces.
 *
 * flipv_mesh_level_set   MeshLevelSet::calculateSignedDistanceField (meshlevelset.cpp:138-150): signed distance on the
 *                        (I+1,J+1,K+1) nodes, Array3d order, into phi_out; closest_out (optional) = closest triangle.
 *                        Inside the exact band (`bandwidth` nodes around each triangle's box) and in sign the result is
 *                        bit-identical to the CPU algorithm; farther out the reference's single breadth-first pass is
 *                        replaced by its fixed point (values mostly smaller, i.e. closer to the true distance; within 1 % otherwise).
 * flipv_add_boundary_mesh FluidSimulation::addBoundary (fluidsimulation.cpp:45-58): level set (band 3), negated if
 *                        `inverted`, min-union into the solid SDF held by the context.
 * flipv_reset_boundary   FluidSimulation::resetBoundary / _initializeBoundary (fluidsimulation.cpp:60-62, 198-239): the
 *                        default box, 3 dx + 1e-6 inside the domain.
 * flipv_add_liquid_mesh  FluidSimulation::addLiquid (fluidsimulation.cpp:64-97): 8 jittered samples per cell, kept where
 *                        the mesh SDF is negative and the solid SDF is not; particles (velocity 0) are APPENDED to the
 *                        context's particles in cell order.  The jitter is the counter-based generator of the host
 *                        mirror's SEED_COUNTER mode (splitmix64 of seed, cell, sample, axis), not libc rand().
 *                        added_out (optional) = number of particles added. */
int flipv_mesh_level_set(flipv_context *ctx, const float *vertices, size_t nvertices, const int *triangles, size_t ntriangles,
                         int bandwidth, float *phi_out, int *closest_out);
int flipv_add_boundary_mesh(flipv_context *ctx, const float *vertices, size_t nvertices, const int *triangles, size_t ntriangles,
                            int inverted);
int flipv_reset_boundary(flipv_context *ctx);
int flipv_add_liquid_mesh(flipv_context *ctx, const float *vertices, size_t nvertices, const int *triangles, size_t ntriangles,
                          unsigned long long seed, size_t *added_out);

/* the seven viscosity control-volume lattices (viscositysolver.cpp:135-178), for parity tests;
 * which: 0 center (I,J,K) 1 U 2 V 3 W 4 edgeU (I,J+1,K+1) 5 edgeV (I+1,J,K+1) 6 edgeW (I+1,J+1,K).
 * Valid after flipv_viscosity_solve. */
int flipv_read_viscosity_volume(flipv_context *ctx, int which, float *out);

/* ---- whole substep / frame ---- */
int flipv_substep(flipv_context *ctx, float dt, flipv_stats *stats);   /* body of the while loop, fluidsimulation.cpp:145-164 */
int flipv_advance(flipv_context *ctx, float dt, flipv_stats *stats);   /* advance(dt), fluidsimulation.cpp:135-168 */

/* ---- measurement ---- */
int flipv_kernel_stats_reset(flipv_context *ctx);
int flipv_kernel_stats_get(flipv_context *ctx, flipv_kernel_stats *out);
int flipv_synchronize(flipv_context *ctx);
/* `reps` back-to-back launches of one SpMV kernel on the current system (after a solve), HIP-event timed.
 * which: 0 pressure, 1 viscosity (the variant the last solve's loop launched), 2 viscosity as the multigrid-preconditioned loop launches it (q = A p and p.q alone).
 * ms_out = average per launch, cells_out = cells swept per launch. */
int flipv_bench_spmv(flipv_context *ctx, int which, int reps, double *ms_out, double *cells_out);
/* device-to-device copy bandwidth (attainable HBM peak): bytes moved (read+write) per second */
int flipv_bench_copy(flipv_context *ctx, size_t bytes, int reps, double *gbps_out);
/* plain streaming kernels on `bytes` of device memory: mode 0 read-only, 1 copy, 2 write-only (one float4 per lane, grid = the array);
 * 3 read-only, 4 copy, 5 five reads : one write -- the pressure SpMV's own byte mix --, 6 ten reads : three writes -- the viscosity SpMV's -- in the tuned form (16 B per lane, grid sized to the
 * CUs, nontemporal loads and stores): the ceiling a stencil kernel of that mix is judged against.  GB/s of bytes moved */
int flipv_bench_stream(flipv_context *ctx, size_t bytes, int reps, int mode, double *gbps_out);

/* ---- multi-GPU: communicator of a slab decomposition (one context per rank) ----
 * RCCL backend: rank 0 calls flipv_comm_get_unique_id (ncclGetUniqueId), the host broadcasts the 128 bytes
 * (torch.distributed / MPI / a file), every rank calls flipv_comm_init_rccl.  Halo planes travel with grouped
 * ncclSend/ncclRecv, the PCG scalars with ncclAllReduce, all on the context's own stream.
 * Local backend: all ranks are contexts of ONE process on one device, each driven by its own host thread; it exists to
 * verify the decomposition against the single-domain result on a one-GPU machine.
 * Limits, checked by the init calls (FLIPV_ERR_INVALID): at most 32 ranks per communicator; along every axis on which a
 * block has neighbours it must be at least ceil(cfl_number) + 3 cells thick (the widest halo; flipv_set_params re-checks it
 * when cfl_number changes, and rejects a cfl_number whose halo exceeds the 8 entries a block context allocates). */
int flipv_comm_unique_id_bytes(void);
int flipv_comm_get_unique_id(void *id_out);
int flipv_comm_init_rccl(flipv_context *ctx, const void *unique_id, int rank, int nranks);   /* slabs along k: grid {1, 1, nranks} */
int flipv_comm_init_local(flipv_context **ctxs, int nranks);
/* process grid dims[3] (ranks along i, j, k), rank = x + dims[0] * (y + dims[1] * z); the context's block must sit at that place
 * of the grid.  Halo copies and reductions are exchanged DIRECTLY with the <= 26 neighbouring blocks (faces, edges and corners: one pack
 * kernel, one group of sends / receives, one unpack kernel per exchange); particles migrate axis by axis (x, y, z) to the adjacent ranks. */
int flipv_comm_init_rccl_grid(flipv_context *ctx, const void *unique_id, int rank, const int *dims);
int flipv_comm_init_local_grid(flipv_context **ctxs, const int *dims);
/* Host-callback backend (FLIPV_VERSION 6): one process per rank like the RCCL backend, but every exchange is staged through host memory and carried by callbacks of the embedding
 * program -- so several ranks may share ONE device.  It rehearses the multi-process path (launcher, rendezvous, per-process block contexts, halos, migration, reductions) on a
 * one-GPU machine; bench.py --comm host and tests/test_gpu_multiprocess.py drive it with torch.distributed over gloo.  Every callback returns 0 on success.
 *   exchange: n operations of ONE group; operation m sends sbytes[m] bytes at sendbuf[m] to rank peer[m] and receives rbytes[m] bytes from it into recvbuf[m] (either may be 0); the
 *             m-th operation of this rank towards a peer pairs with that peer's m-th operation towards this rank (sizes agree); the call returns when every buffer may be reused / read.
 *   allreduce_sum_f64 / _f32: in-place sum over all ranks, the SAME bits on every rank.   barrier: all ranks. */
typedef struct flipv_host_comm {
    void *user;
    int (*exchange)(void *user, int n, const int *peer, const void *const *sendbuf, const size_t *sbytes, void *const *recvbuf, const size_t *rbytes);
    int (*allreduce_sum_f64)(void *user, double *values, size_t n);
    int (*allreduce_sum_f32)(void *user, float *values, size_t n);
    int (*barrier)(void *user);
} flipv_host_comm;
int flipv_comm_init_host_grid(flipv_context *ctx, const flipv_host_comm *callbacks, int rank, const int *dims);
int flipv_comm_finalize(flipv_context *ctx);

#ifdef __cplusplus
}
#endif
#endif /* FLIPV_H */
