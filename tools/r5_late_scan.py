"""Late / resting states: the oracle carries a scene through `nsub` substeps at ITS defaults; at every listed substep the state
(the oracle's particles) is handed to (i) the oracle at 1e-10 / 1e-13 = the solution of the reference's linear systems, (ii) the oracle at its defaults
(how far the reference's own iterate is from its converged answer), (iii) the GPU substep with each parameter set.  Per GPU variant: the end-of-substep
velocity error against (i), the same restricted to faces that are rows of the viscosity system, and the error of each SOLVE in isolation (the oracle's
tight solve applied to the GPU's own inputs of that solve).
    python tools/r5_late_scan.py bunny 64 200 80 60,65,70,75 > profiles/r5/late_scan_bunny64_nu200.log"""
import os, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
from test_oracle_compact_golden import build_host_scene
from flipviscosity3d_amd.capi import Context
from oracle import oraclebind as O

VARIANTS = {
    "default": {},
    "old": dict(viscosity_velocity_tolerance=-1.0, viscosity_mass_scale=-1.0, viscosity_massless_polish=-1),   # round 4's rule
    "critonly": dict(viscosity_mass_scale=-1.0),
    "massonly": dict(viscosity_velocity_tolerance=-1.0),
    "nopolish": dict(viscosity_massless_polish=-1),                                                            # round 5 before the massless clusters were solved apart
    "polishonly": dict(viscosity_velocity_tolerance=-1.0, viscosity_mass_scale=-1.0),                          # round 4's rule + the clusters
    "polishmass": dict(viscosity_velocity_tolerance=-1.0),                                                     # ... + the mass scale, no velocity criterion
    "c1e-4": dict(viscosity_velocity_tolerance=1e-4),
    "eta1e-5": dict(viscosity_velocity_tolerance=1e-5),
    "eta1e-4": dict(viscosity_velocity_tolerance=1e-4),
    "win8": dict(viscosity_velocity_window=8),
    "rounds2": dict(viscosity_stage2_rounds=2),
    "rounds3": dict(viscosity_stage2_rounds=3),
    "vtol1e-7": dict(viscosity_tolerance=1e-7, viscosity_velocity_tolerance=-1.0),
    "vtol1e-8": dict(viscosity_tolerance=1e-8, viscosity_velocity_tolerance=-1.0),
    "tight": dict(precision=1, viscosity_tolerance=1e-9, pressure_rel_tolerance=1e-9, viscosity_max_iterations=5000),
}


def scene(kind, N):
    if kind == "bunny":
        return build_host_scene(N, ("sphere_large.ply", True), ["stanford_bunny.ply"])
    if kind == "honey":
        return build_host_scene(N, None, ["rod.ply", "sheet.ply"])
    raise SystemExit("scene?")


def relmax(A, B, den, masks=None):
    m = 0.0
    for i, (a, b) in enumerate(zip(A, B)):
        d = np.abs(a.astype(np.float64) - b)
        if masks is not None:
            d = d[masks[i]]
        if d.size:
            m = max(m, float(d.max()))
    return m / den


def gpu_substep_by_phase(c, dt):
    """the substep through the per-operator entry points, keeping the inputs and outputs of both solves"""
    out = {}
    c.particle_sdf(); c.advect_velocity_field(); c.body_force(dt)
    out["phi"] = c.grid("LIQUID_PHI")
    out["pre_visc"] = [c.grid(n) for n in "UVW"]
    out["vinfo"] = c.viscosity_solve(dt)
    out["post_visc"] = [c.grid(n) for n in "UVW"]
    c.compute_weights()
    out["w"] = [c.grid("WEIGHT_" + n) for n in "UVW"]
    out["pinfo"] = c.pressure_solve(dt)
    c.apply_pressure(dt)
    out["post_proj"] = [c.grid(n) for n in "UVW"]
    c.extrapolate(); c.constrain()
    out["final"] = [c.grid(n) for n in "UVW"]
    return out


def main():
    kind, N, nu, nsub = sys.argv[1], int(sys.argv[2]), float(sys.argv[3]), int(sys.argv[4])
    at = [int(s) for s in sys.argv[5].split(",")]
    names = sys.argv[6].split(",") if len(sys.argv) > 6 else list(VARIANTS)
    dt = 0.01
    dx, solid, P = scene(kind, N)
    o = O.OracleSim(N, N, N, dx); o.set_solid(solid); o.set_viscosity(nu); o.particles = P
    visc = np.full((N + 1, N + 1, N + 1), nu, np.float32)
    print("# %s %d^3 nu %g (nu dt/dx^2 = %.0f), %d particles" % (kind, N, nu, nu * dt / dx ** 2, len(P)), flush=True)
    for t in range(nsub + 1):
        Pt = o.particles
        if t in at:
            q = O.OracleSim(N, N, N, dx); q.set_solid(solid); q.set_viscosity(nu); q.set_solver_limits(vmaxiter=3000000, vtol=1e-13, ptol=1e-13); q.particles = Pt
            _, vq, pq = q.substep(dt)
            ref = [q.grid(n) for n in "UVW"]; q.close()
            den = max(np.abs(r).max() for r in ref)
            d = O.OracleSim(N, N, N, dx); d.set_solid(solid); d.set_viscosity(nu); d.particles = Pt
            _, vd, pd = d.substep(dt)
            dflt = [d.grid(n) for n in "UVW"]; d.close()
            print("after %3d substeps: max|u| %.3f  reference its %d / %d (1e-13), pressure %d / %d; reference at its defaults vs converged %.2e" %
                  (t, den, vd["iterations"], vq["iterations"], pd["iterations"], pq["iterations"], relmax(dflt, ref, den)), flush=True)
            for name in names:
                c = Context(N, N, N, dx); c.set_solid_sdf(solid); c.set_viscosity(nu)
                if VARIANTS[name]: c.set_params(**VARIANTS[name])
                c.particles = Pt
                g = gpu_substep_by_phase(c, dt)
                # each solve in isolation: the oracle's tight solve on the GPU's own inputs
                p, pi = O.pressure_solve(N, N, N, dx, dt, *g["post_visc"], *g["w"], g["phi"], tol=1e-13, maxiter=100000)
                (pu, pv, pw), _ = O.apply_pressure(N, N, N, dx, dt, p, g["phi"], *g["w"], *g["post_visc"])
                ep = relmax(g["post_proj"], [pu, pv, pw], den)
                v, pr = g["vinfo"], g["pinfo"]
                nbad = sum(int((np.abs(a.astype(np.float64) - b) / den > 1e-4).sum()) for a, b in zip(g["final"], ref))
                print("   %-10s visc its %4d (corr %3d, status %d, prec %d, defect %.1e, step %.1e) pressure its %3d: end of substep %.2e (%d faces > 1e-4) | projection alone %.2e" %
                      (name, v["iterations"], v["correction_iterations"], v["status"], v["preconditioner"], v["defect_residual"] / max(v["rhs_norm"], 1e-300), v["velocity_step"], pr["iterations"],
                       relmax(g["final"], ref, den), nbad, ep), flush=True)
                c.close()
        if t < nsub:
            o.substep(dt)
    o.close()


if __name__ == "__main__":
    main()
