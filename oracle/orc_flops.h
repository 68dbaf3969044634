/*
 * orc_flops.h — instrumented arithmetic type for the CPU oracle (TEST / MEASUREMENT INFRASTRUCTURE, like everything under oracle/).
 *
 * `make liborc_flops.so` compiles the SAME orc_rigid.c as C++ with -DORC_COUNT_FLOPS: `real` becomes a one-word class whose
 * +, -, *, / count themselves, and sqrt / sin / cos / atan2 / pow are counted at their call sites.  The counters are what
 * bench.py reports as F_step, the floating-point operations one env-step of the headline workload takes in this formulation
 * ("Figures for roofline.achieved": achieved_flops = F_step x env-steps/s against the fp32 vector peak).  It is an
 * ALGORITHMIC count of a dense scalar restatement (Cholesky, loop per body): the HIP kernels do the same mathematics with other
 * factorisations and with fused multiply-adds, so their instruction-level count (SQ_INSTS_VALU_* under profiles/) differs; both
 * are reported, each under its own name.
 */
#ifndef ORC_FLOPS_H
#define ORC_FLOPS_H
#ifndef __cplusplus
#error "ORC_COUNT_FLOPS builds compile orc_rigid.c as C++ (see oracle/Makefile)"
#endif
#include <cmath>
#include <cstdint>

struct OrcFlops {
  uint64_t add, mul, div, sqrt_, trans, cmp;
};
extern thread_local OrcFlops orc_flops_tl;

#ifdef ORC_F32
typedef float orc_base;
#else
typedef double orc_base;
#endif

struct real {
  orc_base v;
  real() = default;
  constexpr real(double x) : v((orc_base)x) {}
  constexpr real(float x) : v((orc_base)x) {}
  constexpr real(int x) : v((orc_base)x) {}
  explicit constexpr operator double() const { return (double)v; }
  explicit constexpr operator float() const { return (float)v; }
  explicit constexpr operator int() const { return (int)v; }
  real& operator+=(real b) { orc_flops_tl.add++; v += b.v; return *this; }
  real& operator-=(real b) { orc_flops_tl.add++; v -= b.v; return *this; }
  real& operator*=(real b) { orc_flops_tl.mul++; v *= b.v; return *this; }
  real& operator/=(real b) { orc_flops_tl.div++; v /= b.v; return *this; }
};
static inline real operator+(real a, real b) { orc_flops_tl.add++; return real((double)(orc_base)(a.v + b.v)); }
static inline real operator-(real a, real b) { orc_flops_tl.add++; return real((double)(orc_base)(a.v - b.v)); }
static inline real operator*(real a, real b) { orc_flops_tl.mul++; return real((double)(orc_base)(a.v * b.v)); }
static inline real operator/(real a, real b) { orc_flops_tl.div++; return real((double)(orc_base)(a.v / b.v)); }
static inline real operator-(real a) { return real((double)(-a.v)); }
static inline real operator+(real a) { return a; }
#define ORC_CMP(op) static inline bool operator op(real a, real b) { orc_flops_tl.cmp++; return a.v op b.v; }
ORC_CMP(<) ORC_CMP(>) ORC_CMP(<=) ORC_CMP(>=) ORC_CMP(==) ORC_CMP(!=)
#undef ORC_CMP

/* libm calls are counted where orc_rigid.c makes them (always on doubles cast from `real`) */
static inline double orc_cnt_sqrt(double x) { orc_flops_tl.sqrt_++; return std::sqrt(x); }
static inline double orc_cnt_sin(double x) { orc_flops_tl.trans++; return std::sin(x); }
static inline double orc_cnt_cos(double x) { orc_flops_tl.trans++; return std::cos(x); }
static inline double orc_cnt_atan2(double y, double x) { orc_flops_tl.trans++; return std::atan2(y, x); }
static inline double orc_cnt_pow(double x, double y) { orc_flops_tl.trans++; return std::pow(x, y); }
#define sqrt(x) orc_cnt_sqrt(x)
#define sin(x) orc_cnt_sin(x)
#define cos(x) orc_cnt_cos(x)
#define atan2(y, x) orc_cnt_atan2(y, x)
#define pow(x, y) orc_cnt_pow(x, y)
#endif
