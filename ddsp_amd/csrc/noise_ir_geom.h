// Geometry of the impulse response core.apply_window_to_impulse_response makes (ddsp/core.py:1477-1531): shared by the
// general kernels (filtered_noise.hip) and the constant matrix of filtered_noise_general.hip.
#pragma once
#include <hip/hip_runtime.h>

namespace ddsp {

// ------------------------------------------------------------------------------------
// Geometry of core.apply_window_to_impulse_response (core.py:1477-1531).
// ------------------------------------------------------------------------------------
struct IrGeom {
  int M, L0, ws, padding, half, L;
};
__host__ __device__ inline int ir_first_len(int half) { return half > 2 ? half - 2 : 0; }       // len(ir[L0 - half + 2:])
__host__ __device__ inline IrGeom ir_geom(int M, int window_size) {
  IrGeom g;
  g.M = M;
  g.L0 = 2 * (M - 1);                                     // irfft length (core.py:1559)
  g.ws = (window_size <= 0 || window_size > g.L0) ? g.L0 : window_size;   // :1501-1503
  g.padding = g.L0 - g.ws;
  g.half = (g.ws + 1) / 2;                                // :1509
  // concat(ir[L0 - half + 2:], ir[:half + 1]) (:1520-1527): half - 2 taps and half + 1 taps - but a window of one or two samples
  // (half = 1) starts its first slice at L0 + 1, past the end: python gives an EMPTY slice there, not one of -1 elements, and
  // the filter has two taps (found by tools/fuzz_api_vs_reference.py: the mirror said one)
  g.L = g.padding > 0 ? ir_first_len(g.half) + g.half + 1 : g.L0;
  return g;
}
// tf.signal.hann_window(ws) as TensorFlow computes it (tensorflow/python/ops/signal/window_ops.py, _raised_cosine_window, called
// with periodic=True by core.py:1505): w[i] = 0.5 - 0.5 cos(2 pi i / n), n = window_length + periodic * even - 1 with
// even = 1 - window_length % 2.  "Periodic" therefore holds for EVEN lengths only (n = ws); an ODD length - the constructor's
// default window_size=257 wherever the response is longer than that - gets the SYMMETRIC window, n = ws - 1; a window of one
// sample is [1.0].  (Rounds 1-5 divided by ws whatever its parity: the oracle and the TF stand-in shared that reading, so no
// test could see it; found in round 6 while fixing ADVICE r5's one-sample window.  TensorFlow is not installable in this image: tests/golden/make_golden_tf.py
// regenerates the fixtures under real TF for whoever has it.)
__host__ __device__ inline int hann_denominator(int ws) { return (ws & 1) ? ws - 1 : ws; }      // (0 for ws = 1: the caller's case)
// causal tap index kappa -> zero-phase sample index n and Hann window index (or -1: zero)
__host__ __device__ inline void ir_tap_map(const IrGeom& g, int kappa, int* n, int* widx) {
  if (g.padding > 0) {
    // concat(ir[L0-half+2:], ir[:half+1])                         (core.py:1521-1526)
    const int first = ir_first_len(g.half);
    const int nn = (kappa < first) ? (g.L0 - g.half + 2 + kappa) : (kappa - first);
    // window_zp = concat(window[half:], zeros(padding), window[:half])   (:1510-1512)
    int wi = -1;
    if (nn < g.ws - g.half) wi = g.half + nn;
    else if (nn >= g.L0 - g.half) wi = nn - (g.L0 - g.half);
    *n = nn; *widx = wi;
  } else {
    // fftshift(window) * ir, then fftshift                          (:1514, 1529)
    *n = (kappa + g.L0 / 2) % g.L0;
    *widx = kappa;
  }
}

}  // namespace ddsp
