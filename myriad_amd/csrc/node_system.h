// node_system.h -- neural-ODE dynamics (BASELINE config 5): x' = MLP_theta([x; u]), cost of the TRUE system.
//
// Restates /root/reference/myriad/systems/neural_ode/node_system.py:14-42 (parametrized_dynamics = net.apply(params,
// append(x,u)); the cost stays the true system's, :41-42) with the network of
// /root/reference/myriad/neural_ode/create_node.py:110-117: hk.Linear(h) + sigmoid per hidden layer, then hk.Linear(ns)
// (Haiku Linear: y = x @ w + b, w of shape (in, out)).
//
// Parameter vector (device order, NP doubles, shared or per instance, used IN PLACE through a pointer):
//   w1 [NW][H1] | b1 [H1] | w2 [H1][H2] | b2 [H2] | w3 [H2][NS] | b3 [NS]      (row-major, Haiku's (in, out) layout)
//
// Every lane evaluates the network for ITS collocation point; the weights are wave-uniform, so the compiler reads
// them with scalar loads and the inner products run as v_fma_f64 with a scalar operand -- the fp64 vector rate.
// On gfx950 the fp64 MFMA rate equals the fp64 VALU rate (78.6 TFLOP/s both), so v_mfma_f64_16x16x4 would not be
// faster here; it would only change which lanes hold what.
#pragma once
#include <math.h>
#include "systems_gen.h"

namespace myriad {

template <class True, int H1, int H2, int ID_>
struct SysNODE {
  static constexpr int ID = ID_, NS = True::NS, NU = True::NU, NW = True::NW;
  static constexpr int NP = NW * H1 + H1 + H1 * H2 + H2 + H2 * NS + NS;
  static constexpr int O_W1 = 0, O_B1 = O_W1 + NW * H1, O_W2 = O_B1 + H1, O_B2 = O_W2 + H1 * H2, O_W3 = O_B2 + H2, O_B3 = O_W3 + H2 * NS;
  static constexpr bool COST_DEP_X = True::COST_DEP_X;
  static constexpr bool PARAMS_BY_POINTER = true;
  static constexpr bool HAS_TERMINAL = false;           // NodeSystem is built over systems without a terminal cost
  static constexpr bool TIME_DEP = false;               // ... and with time-independent cost
  static constexpr int T_SLOT = 0;
  MYR_HD static inline double term(const double*, const double*, const double*) { return 0.0; }
  MYR_HD static inline void term_grad(const double*, const double*, const double*, double* gw) { for (int i = 0; i < NW; ++i) gw[i] = 0.0; }
  static constexpr int NNZ2 = 1;                      // no stored second derivatives: hessian() recomputes
  static constexpr const char* NAME = "NODE";

  // The network evaluations are real function calls on the device (MYR_NODE_FN): each is thousands of flops, so the
  // call costs nothing, and the solver kernels stay at a size the register allocator handles (inlined, the wave kernel
  // of this system grew to 40k instructions with ~4k spilled SGPRs).
#ifdef __HIP_DEVICE_COMPILE__
#define MYR_NODE_FN __device__ __attribute__((noinline)) static
#else
#define MYR_NODE_FN MYR_HD static inline
#endif
  MYR_HD static inline double sig(double a) { return 1.0 / (1.0 + exp(-a)); }

  // forward pass keeping the hidden activations
  MYR_HD static inline void fwd(const double* x, const double* u, const double* p, double* h1, double* h2, double* out) {
    double w[NW];
#pragma unroll
    for (int i = 0; i < NS; ++i) w[i] = x[i];
#pragma unroll
    for (int i = 0; i < NU; ++i) w[NS + i] = u[i];
    for (int j = 0; j < H1; ++j) {
      double a = p[O_B1 + j];
#pragma unroll
      for (int i = 0; i < NW; ++i) a += w[i] * p[O_W1 + i * H1 + j];
      h1[j] = sig(a);
    }
    for (int j = 0; j < H2; ++j) {
      double a = p[O_B2 + j];
      for (int i = 0; i < H1; ++i) a += h1[i] * p[O_W2 + i * H2 + j];
      h2[j] = sig(a);
    }
#pragma unroll
    for (int r = 0; r < NS; ++r) {
      double a = p[O_B3 + r];
      for (int i = 0; i < H2; ++i) a += h2[i] * p[O_W3 + i * NS + r];
      out[r] = a;
    }
  }

  MYR_NODE_FN void f(const double* x, const double* u, const double* p, double* fo) {
    double h1[H1], h2[H2];
    fwd(x, u, p, h1, h2, fo);
  }
  // the cost is the TRUE system's (node_system.py:41-42); its parameters are the true defaults
  MYR_HD static inline double g(const double* x, const double* u, const double* p) {
    (void)p;
    double tp[True::NPX];
    True::default_params(tp);
    return True::g(x, u, tp);
  }
  MYR_HD static inline void cost_grad(const double* x, const double* u, const double* p, double* go, double* gw) {
    (void)p;
    double tp[True::NPX];
    True::default_params(tp);
    True::cost_grad(x, u, tp, go, gw);
  }

  // f, A = df/dx, B = df/du (reverse mode: one backward pass per output row), g, dg
  MYR_NODE_FN void lin(const double* x, const double* u, const double* p, double* fo, double* A, double* B, double* go, double* gw) {
    double h1[H1], h2[H2];
    fwd(x, u, p, h1, h2, fo);
    for (int r = 0; r < NS; ++r) {
      double d1[H1];
      for (int i = 0; i < H1; ++i) d1[i] = 0.0;
      for (int j = 0; j < H2; ++j) {
        const double d2 = p[O_W3 + j * NS + r] * h2[j] * (1.0 - h2[j]);
        for (int i = 0; i < H1; ++i) d1[i] += p[O_W2 + i * H2 + j] * d2;
      }
      double gwr[NW];
#pragma unroll
      for (int c = 0; c < NW; ++c) gwr[c] = 0.0;
      for (int i = 0; i < H1; ++i) {
        const double e = d1[i] * h1[i] * (1.0 - h1[i]);
#pragma unroll
        for (int c = 0; c < NW; ++c) gwr[c] += p[O_W1 + c * H1 + i] * e;
      }
#pragma unroll
      for (int c = 0; c < NS; ++c) A[r * NS + c] = gwr[c];
#pragma unroll
      for (int c = 0; c < NU; ++c) B[r * NU + c] = gwr[NS + c];
    }
    cost_grad(x, u, p, go, gw);
  }
  MYR_HD static inline void lin_d2(const double* x, const double* u, const double* p, double* fo, double* A, double* B,
                                   double* go, double* gw, double* D2) {
    lin(x, u, p, fo, A, B, go, gw);
    D2[0] = 0.0;
  }

  // W = wg * d2 g + d2 (mu^T MLP) / dw2   with  d2(mu^T MLP) = W1 (D1 + S1 W2 D2 W2^T S1) W1^T  (w-space, see below)
  //   g2 = W3 mu, D2 = diag(g2 * s''(a2)), g1 = W2 (g2 * s'(a2)), D1 = diag(g1 * s''(a1)), S1 = diag(s'(a1)),
  //   s' = h(1-h), s'' = h(1-h)(1-2h)
  MYR_NODE_FN void hessian(const double* x, const double* u, const double* p, const double* D2unused,
                                    const double* mu, double wg, double* W) {
    (void)D2unused;
    double h1[H1], h2[H2], out[NS];
    fwd(x, u, p, h1, h2, out);
    double e2[H2], c2[H2];                       // e2 = g2 * s'(a2), c2 = g2 * s''(a2)
    for (int j = 0; j < H2; ++j) {
      double g2 = 0.0;
#pragma unroll
      for (int r = 0; r < NS; ++r) g2 += p[O_W3 + j * NS + r] * mu[r];
      const double sp = h2[j] * (1.0 - h2[j]);
      e2[j] = g2 * sp;
      c2[j] = g2 * sp * (1.0 - 2.0 * h2[j]);
    }
    // V = S1 W1^T  (H1 x NW): d a1 / d w scaled by s'(a1);  then  M = W2^T V  (H2 x NW)
    double Wacc[NW * NW];
#pragma unroll
    for (int i = 0; i < NW * NW; ++i) Wacc[i] = 0.0;
    for (int i = 0; i < H1; ++i) {               // D1 term
      double g1 = 0.0;
      for (int j = 0; j < H2; ++j) g1 += p[O_W2 + i * H2 + j] * e2[j];
      const double sp = h1[i] * (1.0 - h1[i]);
      const double c1 = g1 * sp * (1.0 - 2.0 * h1[i]);
#pragma unroll
      for (int a = 0; a < NW; ++a)
#pragma unroll
        for (int b = a; b < NW; ++b) Wacc[a * NW + b] += c1 * p[O_W1 + a * H1 + i] * p[O_W1 + b * H1 + i];
    }
    for (int j = 0; j < H2; ++j) {               // D2 term: column j of M = sum_i W2[i][j] s'(a1_i) W1[:, i]
      double m[NW];
#pragma unroll
      for (int a = 0; a < NW; ++a) m[a] = 0.0;
      for (int i = 0; i < H1; ++i) {
        const double t = p[O_W2 + i * H2 + j] * h1[i] * (1.0 - h1[i]);
#pragma unroll
        for (int a = 0; a < NW; ++a) m[a] += t * p[O_W1 + a * H1 + i];
      }
#pragma unroll
      for (int a = 0; a < NW; ++a)
#pragma unroll
        for (int b = a; b < NW; ++b) Wacc[a * NW + b] += c2[j] * m[a] * m[b];
    }
    // true cost's second derivative
    double tp[True::NPX], tf[NS], tA[NS * NS], tB[NS * NU], tg, tgw[NW], tD2[True::NNZ2], Wg[NW * NW], zero[NS];
    True::default_params(tp);
#pragma unroll
    for (int r = 0; r < NS; ++r) zero[r] = 0.0;
    True::lin_d2(x, u, tp, tf, tA, tB, &tg, tgw, tD2);
    True::contract(tD2, zero, wg, Wg);
#pragma unroll
    for (int a = 0; a < NW; ++a)
#pragma unroll
      for (int b = a; b < NW; ++b) { const double v = Wacc[a * NW + b] + Wg[a * NW + b]; W[a * NW + b] = v; W[b * NW + a] = v; }
  }
  // the same with the network part already contracted (node_mfma.h, MODE 2): Wm = upper triangle of
  // sum_r mu_r d2 MLP_r / dw2, row by row; adds the true cost's second derivative
  MYR_HD static inline void hessian_packed(const double* x, const double* u, const double* Wm, double wg, double* W) {
    double tp[True::NPX], tf[NS], tA[NS * NS], tB[NS * NU], tg, tgw[NW], tD2[True::NNZ2], Wg[NW * NW], zero[NS];
    True::default_params(tp);
#pragma unroll
    for (int r = 0; r < NS; ++r) zero[r] = 0.0;
    True::lin_d2(x, u, tp, tf, tA, tB, &tg, tgw, tD2);
    True::contract(tD2, zero, wg, Wg);
    int e = 0;
#pragma unroll
    for (int a = 0; a < NW; ++a)
#pragma unroll
      for (int b = a; b < NW; ++b, ++e) { const double v = Wm[e] + Wg[a * NW + b]; W[a * NW + b] = v; W[b * NW + a] = v; }
  }
  MYR_HD static inline void default_params(double* p) { (void)p; }
};

using SysNODE_CARTPOLE = SysNODE<SysCARTPOLE, 64, 64, 4>;     // BASELINE config 5: hidden_layers = (64, 64)

}  // namespace myriad
