// rollout.h -- true-dynamics rollout of [x; integral of g] under given controls (K8 in DESIGN.md).
// Replaces /root/reference/myriad/utils.py:258-298 (get_state_trajectory_and_cost) -> utils.integrate :22-73,
// with the reference's fixed-step rules (:31-54): Euler, Heun, "midpoint" (a full Euler predictor, quirk Q11),
// RK4 (u[2i], u[2i+1] for k2 AND k3, u[2i+2]; quirk Q5).  Control indices beyond the array clamp to the last
// row, as jnp gathers do (quirk Q6).  One trajectory per lane; host/device shared.
#pragma once
#include "systems_gen.h"

namespace myriad {

template <class Sys>
struct Rollout {
  static constexpr int NS = Sys::NS, NU = Sys::NU;

  MYR_HD static inline void aug(const double* x, const double* u, const double* p, double* dx, double* dc, double t = 0.0) {
    set_time<Sys>(p, t);
    Sys::f(x, u, p, dx);
    *dc = Sys::g(x, u, p);
  }

  // us: [u_rows][NU] for this trajectory; xs: [num_steps+1][NS] or null; returns integrated cost
  MYR_HD static double run(int method, int num_steps, double h, int u_rows, const double* x0, const double* us,
                           const double* p, double* xs) {
    double x[NS], c = 0.0;
#pragma unroll
    for (int i = 0; i < NS; ++i) x[i] = x0[i];
    if (xs) {
#pragma unroll
      for (int i = 0; i < NS; ++i) xs[i] = x[i];
    }
    auto U = [&](int i) { return us + (long)(i < u_rows ? i : u_rows - 1) * NU; };
    for (int s = 0; s < num_steps; ++s) {
      double k1[NS], c1, k2[NS], c2, xt[NS];
      const double t = h * s;                  // ts[idx] of linspace(0, T, num_steps+1) (utils.py:273-281); stage times as utils.py:31-54
      if (method == 0) {                       // Euler: x + h f(x, u_i)
        aug(x, U(s), p, k1, &c1, t);
#pragma unroll
        for (int i = 0; i < NS; ++i) x[i] += h * k1[i];
        c += h * c1;
      } else if (method == 1) {                // Heun: x + h/2 (k1 + k2), k2 at (x + h k1, u_{i+1})
        aug(x, U(s), p, k1, &c1, t);
#pragma unroll
        for (int i = 0; i < NS; ++i) xt[i] = x[i] + h * k1[i];
        aug(xt, U(s + 1), p, k2, &c2, t + h);
#pragma unroll
        for (int i = 0; i < NS; ++i) x[i] += 0.5 * h * (k1[i] + k2[i]);
        c += 0.5 * h * (c1 + c2);
      } else if (method == 2) {                // reference "midpoint": x + h f(x + h f(x,u_i), (u_i+u_{i+1})/2)
        aug(x, U(s), p, k1, &c1, t);
        double um[NU];
#pragma unroll
        for (int i = 0; i < NS; ++i) xt[i] = x[i] + h * k1[i];
#pragma unroll
        for (int i = 0; i < NU; ++i) um[i] = 0.5 * (U(s)[i] + U(s + 1)[i]);
        aug(xt, um, p, k2, &c2, t + 0.5 * h);
#pragma unroll
        for (int i = 0; i < NS; ++i) x[i] += h * k2[i];
        c += h * c2;
      } else {                                 // RK4 with controls u[2s], u[2s+1] (k2 and k3), u[2s+2]
        double k3[NS], c3, k4[NS], c4;
        aug(x, U(2 * s), p, k1, &c1, t);
#pragma unroll
        for (int i = 0; i < NS; ++i) xt[i] = x[i] + 0.5 * h * k1[i];
        aug(xt, U(2 * s + 1), p, k2, &c2, t + 0.5 * h);
#pragma unroll
        for (int i = 0; i < NS; ++i) xt[i] = x[i] + 0.5 * h * k2[i];
        aug(xt, U(2 * s + 1), p, k3, &c3, t + 0.5 * h);
#pragma unroll
        for (int i = 0; i < NS; ++i) xt[i] = x[i] + h * k3[i];
        aug(xt, U(2 * s + 2), p, k4, &c4, t + h);
#pragma unroll
        for (int i = 0; i < NS; ++i) x[i] += h / 6.0 * (k1[i] + 2.0 * k2[i] + 2.0 * k3[i] + k4[i]);
        c += h / 6.0 * (c1 + 2.0 * c2 + 2.0 * c3 + c4);
      }
      if (xs) {
#pragma unroll
        for (int i = 0; i < NS; ++i) xs[(long)(s + 1) * NS + i] = x[i];
      }
    }
    if constexpr (Sys::HAS_TERMINAL) c += Sys::term(x, us + (long)(u_rows - 1) * NU, p);   // utils.py:295-296: terminal_cost_fn(x_end, us[-1])
    return c;
  }
};

}  // namespace myriad
