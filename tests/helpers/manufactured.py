"""Manufactured solutions for the full-size solver checks (tests/test_gpu_fullsize.py, bench.py).

A 12M-dof direct solve is out of reach of the CPU checker, so the solver term at BASELINE's 4M-triangle sizes is
measured the other way round: pick a smooth displacement field u* that vanishes on the fixed dofs, let the device
evaluate b = K u* with double-double products and sums (femshell_residual with zero loads returns -K u*), hand b back
as the load vector and solve.  b is rounded to double on the way (eps |b_i| per entry); the solution of K u = fl(b) is
u* + delta with K delta = fl(b) - K u*, a right-hand side the double-double residual gives exactly, so delta (a
correction of relative size ~1e-13) is obtained with one more solve and the reference is u* + delta.

Test infrastructure only (reads nothing of the oracle; the product is driven through its C ABI).
"""
import numpy as np


def smooth_field(m, kind):
    """u*[n_nodes, 6]: smooth, of the magnitude of the load case's displacements, zero on every fixed dof."""
    x = np.asarray(m.xyz, dtype=np.float64)
    u = np.zeros((len(x), 6))
    if kind == "panel":
        # 10 x 10 plate in the x-y plane, all edges simply supported (u, v, w fixed): Navier-type deflection with the
        # Kirchhoff rotations of that deflection, in-plane sines a fiftieth of it
        L = float(x[:, 0].max() - x[:, 0].min())
        X, Y = (x[:, 0] - x[:, 0].min()) / L, (x[:, 1] - x[:, 1].min()) / L
        W = 0.1
        u[:, 2] = W * np.sin(np.pi * X) * np.sin(np.pi * Y)
        u[:, 3] = W * np.pi / L * np.sin(np.pi * X) * np.cos(np.pi * Y)      # theta_x = dw/dy
        u[:, 4] = -W * np.pi / L * np.cos(np.pi * X) * np.sin(np.pi * Y)     # theta_y = -dw/dx
        u[:, 0] = W / 50 * np.sin(2 * np.pi * X) * np.sin(np.pi * Y)
        u[:, 1] = W / 50 * np.sin(np.pi * X) * np.sin(2 * np.pi * Y)
        u[:, 5] = W / 500 * np.sin(np.pi * X) * np.sin(np.pi * Y)
    elif kind == "cylinder":
        # axis z, both end rings fixed in u, v, w: ovalisation cos(2 theta) sin(pi z / L) of the radius with the
        # tangential displacement that keeps the circumference (v_t = -W/2 sin(2 theta)), small axial part, rotations
        # of the radial deflection
        R = float(np.hypot(x[:, 0], x[:, 1]).mean())
        L = float(x[:, 2].max() - x[:, 2].min())
        th = np.arctan2(x[:, 1], x[:, 0])
        Z = (x[:, 2] - x[:, 2].min()) / L
        W = 1e-5 * R
        s = np.sin(np.pi * Z)
        wr = W * np.cos(2 * th) * s
        vt = -0.5 * W * np.sin(2 * th) * s
        er = np.stack([np.cos(th), np.sin(th), np.zeros_like(th)], axis=1)
        et = np.stack([-np.sin(th), np.cos(th), np.zeros_like(th)], axis=1)
        u[:, 0:3] = wr[:, None] * er + vt[:, None] * et
        u[:, 2] += W / 40 * np.cos(2 * th) * np.sin(2 * np.pi * Z)
        # rotation vector of the mid-surface normal: about e_t by -dw/dz, about e_z by (dw/dtheta - v_t) / R
        dwdz = W * np.cos(2 * th) * np.pi / L * np.cos(np.pi * Z)
        dwdt = -2 * W * np.sin(2 * th) * s
        u[:, 3:6] = (-dwdz)[:, None] * et + (((dwdt - vt) / R))[:, None] * np.array([0.0, 0.0, 1.0])
    elif kind == "flap":
        # BASELINE configs[4]: 0.1 x 1 flap in the x-z plane (normal y), bottom edge z = 0 fixed in u, v, w, loaded in its own
        # plane (fluid_solver.cpp:192: f_x on the left edge): a cantilever-type deflection along x with the axial displacement
        # of the cross-sections' rotation and that rotation as the nodal rotation about the normal.  No out-of-plane part: the
        # coupled case never bends the flap out of its plane, and on its 0.0002 x 0.001 cells FP64 does not resolve those modes
        # next to the in-plane ones (measured: a manufactured deflection along y of 2 % of the in-plane one leaves the
        # correction solve at a relative residual of 3e-6 after 2000 iterations, tools/lab/flap_manufactured_probe.py)
        L = float(x[:, 2].max() - x[:, 2].min())
        Z = (x[:, 2] - x[:, 2].min()) / L
        xm = 0.5 * float(x[:, 0].max() + x[:, 0].min())
        W = 1.0
        shape, slope = 0.5 * Z ** 2 * (3.0 - Z), 1.5 * Z * (2.0 - Z) / L
        u[:, 0] = W * shape
        u[:, 2] = -W * slope * (x[:, 0] - xm)
        u[:, 4] = W * slope                                   # rotation about y (the normal)
    else:
        raise ValueError(kind)
    mask = m.dirichlet_mask()
    for v in range(6):
        u[(mask >> v) & 1 == 1, v] = 0.0
    return u


def rhs_of(fs, u_star):
    """b = K u* from the device's double-double residual (zero loads: r = 0 - K u*), rounded to double."""
    n = u_star.shape[0]
    fs.set_loads(np.zeros((n, 6)))
    return -fs.residual(u_star).reshape(n, 6)


def rounding_correction(fs, u_star, rtol=1e-6, max_it=2000):
    """delta with K delta = fl(b) - K u* (the loads must be fl(b) = rhs_of(fs, u_star)); the exact solution of the
    system the solver gets is u* + delta.  Leaves the correction's right-hand side as the loads: set them again."""
    n = u_star.shape[0]
    rho = fs.residual(u_star).reshape(n, 6)  # fl(b) - K u*, evaluated in double-double
    if not np.any(rho):
        return np.zeros_like(u_star), 0.0
    fs.set_loads(rho)
    d, info = fs.solve(rtol=rtol, max_it=max_it)
    assert info["converged"] == 1, info
    return d.reshape(n, 6), float(np.linalg.norm(rho))
