// ---------------------------------------------------------------------------
// ref_physics.hpp — TEST INFRASTRUCTURE (builds only where /root/reference is present).
//
// The remaining (dispersion x solver x equilibrium) combinations that the reference's
// own tests exercise (graph_tests/solver_test.cpp, graph_tests/physics_test.cpp), restated
// against the reference node API like ref_builders.hpp: analytic equilibria
// (equilibrium.hpp:476-1106), the scalar dispersion relations (dispersion.hpp:400-900),
// and the rk2 / split_simplextic / adaptive_rk4 integrators (solver.hpp:551-670,
// 877-1170).  The graphs are evaluated by the tape interpreter of ref_builders.hpp and
// exported as GFIR for the HIP backend.
// ---------------------------------------------------------------------------
#ifndef ref_physics_hpp
#define ref_physics_hpp

#include "ref_builders.hpp"

// ---------------------------------------------------------------------------
// Analytic equilibria.  `kind` selects the profile set:
//   no_magnetic_field  ne = ni = 1e19 (0.1 x + 1), te = ti = 1000, B = 0            (:482-583)
//   slab               ne = ni = 1e19, te = ti = 1000, B = (0, 0, 0.1 x + 1)        (:611-707)
//   slab_density       ne = ni = 1e19 (0.1 x + 1), te = ti = 1000, B = (0, 0, 1)    (:735-836)
//   slab_field         ne = ni = 1e19 (0.01 x + 1), te = ti = 2000 (0.01 x + 1),
//                      B = (0, 0, 0.01 x + 1)                                       (:864-964)
//   gaussian_density   ne = ni = 1e19 exp((x^2 + y^2)/-0.2), te = ti = 1000,
//                      B = (1, 0, 0)                                                (:991-1090)
// ---------------------------------------------------------------------------
enum class analytic_kind {no_magnetic_field, slab, slab_density, slab_field, gaussian_density};

template<typename T>
struct analytic_equilibrium : public equilibrium_base<T> {
    const analytic_kind kind;
    explicit analytic_equilibrium(const analytic_kind k) : kind(k) {}

    static leaf<T> c(const double value) { return graph::constant<T> (static_cast<T> (value)); }

    leaf<T> linear(const double amplitude, const double slope, leaf<T> x) const {
        return c(amplitude)*(c(slope)*x + graph::one<T> ());
    }

    leaf<T> get_electron_density(leaf<T> x, leaf<T> y, leaf<T> z) override {
        (void)z;
        switch (kind) {
            case analytic_kind::slab: return c(1.0E19);
            case analytic_kind::no_magnetic_field:
            case analytic_kind::slab_density: return linear(1.0E19, 0.1, x);
            case analytic_kind::slab_field: return linear(1.0E19, 0.01, x);
            default: return c(1.0E19)*graph::exp((x*x + y*y)/c(-0.2));
        }
    }
    leaf<T> get_ion_density(leaf<T> x, leaf<T> y, leaf<T> z) override {
        return get_electron_density(x, y, z);
    }
    leaf<T> get_electron_temperature(leaf<T> x, leaf<T> y, leaf<T> z) override {
        (void)y; (void)z;
        return kind == analytic_kind::slab_field ? linear(2000.0, 0.01, x) : c(1000.0);
    }
    leaf<T> get_ion_temperature(leaf<T> x, leaf<T> y, leaf<T> z) override {
        return get_electron_temperature(x, y, z);
    }
    vec3<T> get_magnetic_field(leaf<T> x, leaf<T> y, leaf<T> z) override {
        (void)y; (void)z;
        auto zero = graph::zero<T> ();
        auto one = graph::one<T> ();
        switch (kind) {
            case analytic_kind::no_magnetic_field: return graph::vector(zero, zero, zero);
            case analytic_kind::slab: return graph::vector(0.0, 0.0, 0.1*x + 1.0);
            case analytic_kind::slab_density: return graph::vector(zero, zero, one);
            case analytic_kind::slab_field: return graph::vector(0.0, 0.0, 0.01*x + 1.0);
            default: return graph::vector(one, zero, zero);
        }
    }
};

// ---------------------------------------------------------------------------
// Dispersion relations (same signature as cold_plasma_D / ordinary_wave_D).
// ---------------------------------------------------------------------------
template<typename T>
struct physics_constants {                                                  // dispersion.hpp:490-503
    T epsilon0, mu0, q, me, c;
    physics_constants() {
        epsilon0 = 8.8541878138E-12;
        mu0 = M_PI*4.0E-7;
        q = 1.602176634E-19;
        me = 9.1093837015E-31;
        c = static_cast<T> (1.0)/std::sqrt(epsilon0*mu0);
    }
    leaf<T> plasma_frequency(leaf<T> n, const T charge, const T m) const { // :326-332
        return n*charge*charge/(epsilon0*m*c*c);
    }
    leaf<T> cyclotron_frequency(const T charge, leaf<T> b, const T m) const {   // :348-353
        return charge*b/(m*c);
    }
};

//  dispersion::simple, dispersion.hpp:451-485.
template<typename T>
leaf<T> simple_D(leaf<T> w, vec3<T> k, leaf<T>, leaf<T>, leaf<T>, equilibrium_base<T> &, std::vector<leaf<T>> *) {
    const T c = 1.0;
    auto npar2 = k->get_z()*k->get_z()*c*c/(w*w);
    auto nperp2 = (k->get_x()*k->get_x() + k->get_y()*k->get_y())*c*c/(w*w);
    return npar2 + nperp2 - c;
}

//  dispersion::gaussian_well, dispersion.hpp:684-717.
template<typename T>
leaf<T> gaussian_well_D(leaf<T> w, vec3<T> k, leaf<T> x, leaf<T> y, leaf<T>, equilibrium_base<T> &, std::vector<leaf<T>> *) {
    const T c = 1.0;
    auto well = c - 0.5*graph::exp(-(x*x + y*y)/0.1);
    auto npar2 = k->get_z()*k->get_z()*c*c/(w*w);
    auto nperp2 = (k->get_x()*k->get_x() + k->get_y()*k->get_y())*c*c/(w*w);
    return npar2 + nperp2 - well;
}

//  k_parallel^2 of bohm_gross / acoustic_wave (dispersion.hpp:551-558, 664-671).
template<typename T>
leaf<T> parallel_k2(vec3<T> k, vec3<T> b_vec) {
    if (b_vec->length()->is_match(graph::zero<T> ())) {
        return k->dot(k);
    }
    auto kpara = b_vec->unit()->dot(k);
    return kpara*kpara;
}

//  dispersion::bohm_gross, dispersion.hpp:512-565.
template<typename T>
leaf<T> bohm_gross_D(leaf<T> w, vec3<T> k, leaf<T> x, leaf<T> y, leaf<T> z, equilibrium_base<T> &eq, std::vector<leaf<T>> *) {
    const physics_constants<T> p;
    auto wpe2 = p.plasma_frequency(eq.get_electron_density(x, y, z), p.q, p.me);
    auto te = eq.get_electron_temperature(x, y, z);
    auto vterm2 = static_cast<T> (2.0)*p.q*te/(p.me*p.c*p.c);
    auto kpara2 = parallel_k2<T> (k, eq.get_magnetic_field(x, y, z));
    return wpe2 + 3.0/2.0*kpara2*vterm2 - w*w;
}

//  dispersion::light_wave, dispersion.hpp:575-617.
template<typename T>
leaf<T> light_wave_D(leaf<T> w, vec3<T> k, leaf<T> x, leaf<T> y, leaf<T> z, equilibrium_base<T> &eq, std::vector<leaf<T>> *) {
    const physics_constants<T> p;
    auto wpe2 = p.plasma_frequency(eq.get_electron_density(x, y, z), p.q, p.me);
    return wpe2 + k->dot(k) - w*w;
}

//  dispersion::acoustic_wave, dispersion.hpp:627-674.
template<typename T>
leaf<T> acoustic_wave_D(leaf<T> w, vec3<T> k, leaf<T> x, leaf<T> y, leaf<T> z, equilibrium_base<T> &eq, std::vector<leaf<T>> *) {
    const physics_constants<T> p;
    const T mi = 3.34449469E-27;
    auto te = eq.get_electron_temperature(x, y, z);
    auto ti = eq.get_ion_temperature(x, y, z);
    const T gamma = 3.0;
    auto vs2 = (p.q*te + gamma*p.q*ti)/(mi*p.c*p.c);
    auto kpara2 = parallel_k2<T> (k, eq.get_magnetic_field(x, y, z));
    return kpara2*vs2 - w*w;
}

//  dispersion::extra_ordinary_wave, dispersion.hpp:838-894.
template<typename T>
leaf<T> extra_ordinary_wave_D(leaf<T> w, vec3<T> k, leaf<T> x, leaf<T> y, leaf<T> z, equilibrium_base<T> &eq, std::vector<leaf<T>> *) {
    const physics_constants<T> p;
    auto wpe2 = p.plasma_frequency(eq.get_electron_density(x, y, z), p.q, p.me);
    auto b_vec = eq.get_magnetic_field(x, y, z);
    auto wec = p.cyclotron_frequency(-p.q, b_vec->length(), p.me);
    auto n = k/w;
    auto nperp = b_vec->unit()->cross(n);
    auto nperp2 = nperp->dot(nperp);
    auto wh = wpe2 + wec*wec;
    auto w2 = w*w;
    return 1.0 - wpe2/(w2)*(w2 - wpe2)/(w2 - wh) - nperp2;
}

// ---------------------------------------------------------------------------
// Integrators: the `solver_kernel` item of solver_interface::compile (solver.hpp:303-349)
// for rk2 (:593-660) and split_simplextic (:1031-1160).  rk4 is make_solver_kernel.
// ---------------------------------------------------------------------------
template<typename T>
work_item<T> ray_item(const ray_variables<T> &v, leaf<T> residual, leaf<T> kx_next, leaf<T> ky_next,
                      leaf<T> kz_next, leaf<T> x_next, leaf<T> y_next, leaf<T> z_next, leaf<T> t_next) {
    return work_item<T> (v.inputs(), {residual},
                         {{kx_next, v.kx}, {ky_next, v.ky}, {kz_next, v.kz},
                          {x_next, v.x}, {y_next, v.y}, {z_next, v.z}, {t_next, v.t}});
}

template<typename T>
work_item<T> make_rk2_kernel(const ray_variables<T> &v, equilibrium_base<T> &eq, const T dt_value,
                             dispersion_interface<T> &D) {
    auto dt = graph::constant<T> (dt_value);
    auto kx1 = dt*D.dkxdt, ky1 = dt*D.dkydt, kz1 = dt*D.dkzdt;
    auto x1 = dt*D.dxdt, y1 = dt*D.dydt, z1 = dt*D.dzdt;
    dispersion_interface<T> D2(v.w,
                               graph::pseudo_variable(v.kx + kx1),
                               graph::pseudo_variable(v.ky + ky1),
                               graph::pseudo_variable(v.kz + kz1),
                               graph::pseudo_variable(v.x + x1),
                               graph::pseudo_variable(v.y + y1),
                               graph::pseudo_variable(v.z + z1), eq, D.function);
    auto kx2 = dt*D2.dkxdt, ky2 = dt*D2.dkydt, kz2 = dt*D2.dkzdt;
    auto x2 = dt*D2.dxdt, y2 = dt*D2.dydt, z2 = dt*D2.dzdt;
    return ray_item<T> (v, D.D*D.D,
                        v.kx + (kx1 + kx2)/2.0, v.ky + (ky1 + ky2)/2.0, v.kz + (kz1 + kz2)/2.0,
                        v.x + (x1 + x2)/2.0, v.y + (y1 + y2)/2.0, v.z + (z1 + z2)/2.0, v.t + dt);
}

template<typename T>
work_item<T> make_split_simplextic_kernel(const ray_variables<T> &v, equilibrium_base<T> &eq, const T dt_value,
                                          dispersion_interface<T> &D) {
    auto dt = graph::constant<T> (dt_value);
    auto t_next = v.t + dt;
    auto x1 = v.x + dt*D.dxdt/2.0;
    auto y1 = v.y + dt*D.dydt/2.0;
    auto z1 = v.z + dt*D.dzdt/2.0;
    dispersion_interface<T> D2(v.w,
                               graph::pseudo_variable(v.kx), graph::pseudo_variable(v.ky), graph::pseudo_variable(v.kz),
                               graph::pseudo_variable(x1), graph::pseudo_variable(y1), graph::pseudo_variable(z1),
                               eq, D.function);
    auto kx_next = v.kx + dt*D2.dkxdt;
    auto ky_next = v.ky + dt*D2.dkydt;
    auto kz_next = v.kz + dt*D2.dkzdt;
    dispersion_interface<T> D3(v.w,
                               graph::pseudo_variable(kx_next), graph::pseudo_variable(ky_next),
                               graph::pseudo_variable(kz_next),
                               graph::pseudo_variable(x1), graph::pseudo_variable(y1), graph::pseudo_variable(z1),
                               eq, D.function);
    return ray_item<T> (v, D.D*D.D, kx_next, ky_next, kz_next,
                        x1 + dt*D3.dxdt/2.0, y1 + dt*D3.dydt/2.0, z1 + dt*D3.dzdt/2.0, t_next);
}

#endif /* ref_physics_hpp */
