"""Direct numpy evaluation of the VMEC equilibrium from the spline tables of the reference-held
graph_tests/vmec.nc (tests/golden/vmec_tables.npz) — TEST INFRASTRUCTURE, independent of the graph layer:
cubic splines in t = (s - offset)/ds per flux-surface interval, plain Fourier sums with ANALYTIC derivatives
(no df()), the covariant basis, the Jacobian, the contravariant basis, B and the profiles
(equilibrium.hpp:1920-2013, :2073-2150).  chi is evaluated the way the reference does it — on
s_norm_f = (s - sminf)/ds passed where a flux label is expected (:2133 -> :2036-2046), table index and all.
`modes` truncates the Fourier sums to the first modes of the file, as oracle/ref_vmec.cpp does."""
import numpy as np


def vmec_field(tables, s, u, v, modes=None):
    sminh, sminf, ds = float(tables["sminh"]), float(tables["sminf"]), float(tables["ds"])

    def spline(prefix, mode, x, offset):
        """value and d/dx of the cubic spline (piecewise.hpp:26-65 index: clamp, truncate)."""
        c = [tables["%s_c%d" % (prefix, k)] for k in range(4)]
        if mode is not None:
            c = [ck[mode] for ck in c]
        t = (x - offset)/ds
        index = np.clip(t, 0.0, c[0].size - 1.0).astype(np.int64)
        c0, c1, c2, c3 = (ck[index] for ck in c)
        return c0 + t*(c1 + t*(c2 + t*c3)), (c1 + t*(2.0*c2 + 3.0*t*c3))/ds

    r = r_s = r_u = r_v = z = z_s = z_u = z_v = l_u = l_v = np.zeros_like(s)
    count = tables["xm"].size if modes is None else min(int(modes), tables["xm"].size)
    for i in range(count):
        m, n = tables["xm"][i], tables["xn"][i]
        angle = m*u - n*v
        sin, cos = np.sin(angle), np.cos(angle)
        rmnc, rmnc_s = spline("rmnc", i, s, sminf)
        zmns, zmns_s = spline("zmns", i, s, sminf)
        lmns, _ = spline("lmns", i, s, sminh)
        r, r_s, r_u, r_v = r + rmnc*cos, r_s + rmnc_s*cos, r_u - m*rmnc*sin, r_v + n*rmnc*sin
        z, z_s, z_u, z_v = z + zmns*sin, z_s + zmns_s*sin, z_u + m*zmns*cos, z_v - n*zmns*cos
        l_u, l_v = l_u + m*lmns*cos, l_v - n*lmns*cos

    def rotate(a, b, c):                                             # get_esubs/u/v: (a, b, c) in (R, phi, Z) components
        return np.stack([np.cos(v)*a - np.sin(v)*b, np.sin(v)*a + np.cos(v)*b, c])

    zero = np.zeros_like(s)
    esubs, esubu, esubv = rotate(r_s, zero, z_s), rotate(r_u, zero, z_u), rotate(r_v, r, z_v)
    jacobian = np.sum(esubs*np.cross(esubu, esubv, axis=0), axis=0)
    s_norm_f = (s - sminf)/ds
    _, chi_x = spline("chi", None, s_norm_f, sminf)
    chi_s = chi_x/ds                                                 # d s_norm_f / ds
    phip = float(tables["signj"])*float(tables["dphi"])
    b = ((chi_s - phip*l_v)*esubu + phip*(1.0 + l_u)*esubv)/jacobian
    profile = (1.0 - np.sqrt(s*s)**1.5)**2
    return dict(b=b, r=r, z=z, jacobian=jacobian, profile=profile, esubs=esubs, esubu=esubu, esubv=esubv,
                esups=np.cross(esubu, esubv, axis=0)/jacobian, esupu=np.cross(esubv, esubs, axis=0)/jacobian,
                esupv=np.cross(esubs, esubu, axis=0)/jacobian)
