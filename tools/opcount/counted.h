// counted.h -- an arithmetic type that counts floating-point operations, used to compile oracle/dial_oracle.c as C++
// (tools/opcount/count_flops.py).  TEST / MEASUREMENT INFRASTRUCTURE: gives the FLOP count of the dense reference
// formulation of one env.step (SURVEY 8d asked for counted instead of estimated work).  mul, add/sub, fma-able pairs
// are counted separately; divisions, square roots and transcendental calls are counted as such.
#pragma once
#include <cmath>
#include <cstdlib>
struct OpCounts { unsigned long long add, mul, div, sqrt_, trans, cmp; };
extern thread_local OpCounts g_ops;
struct CReal {
  double v;
  CReal() : v(0) {}
  CReal(double x) : v(x) {}
  CReal(float x) : v(x) {}
  CReal(int x) : v(x) {}
  explicit operator double() const { return v; }
  explicit operator float() const { return (float)v; }
  explicit operator int() const { return (int)v; }
  CReal& operator+=(const CReal& o) { g_ops.add++; v += o.v; return *this; }
  CReal& operator-=(const CReal& o) { g_ops.add++; v -= o.v; return *this; }
  CReal& operator*=(const CReal& o) { g_ops.mul++; v *= o.v; return *this; }
  CReal& operator/=(const CReal& o) { g_ops.div++; v /= o.v; return *this; }
};
inline CReal operator+(const CReal& a, const CReal& b) { g_ops.add++; return CReal(a.v + b.v); }
inline CReal operator-(const CReal& a, const CReal& b) { g_ops.add++; return CReal(a.v - b.v); }
inline CReal operator*(const CReal& a, const CReal& b) { g_ops.mul++; return CReal(a.v * b.v); }
inline CReal operator/(const CReal& a, const CReal& b) { g_ops.div++; return CReal(a.v / b.v); }
inline CReal operator-(const CReal& a) { return CReal(-a.v); }
inline bool operator<(const CReal& a, const CReal& b) { g_ops.cmp++; return a.v < b.v; }
inline bool operator>(const CReal& a, const CReal& b) { g_ops.cmp++; return a.v > b.v; }
inline bool operator<=(const CReal& a, const CReal& b) { g_ops.cmp++; return a.v <= b.v; }
inline bool operator>=(const CReal& a, const CReal& b) { g_ops.cmp++; return a.v >= b.v; }
inline bool operator==(const CReal& a, const CReal& b) { g_ops.cmp++; return a.v == b.v; }
inline bool operator!=(const CReal& a, const CReal& b) { g_ops.cmp++; return a.v != b.v; }
#define CR_MIXED(T)                                                                                              \
  inline CReal operator+(const CReal& a, T b) { return a + CReal(b); } inline CReal operator+(T a, const CReal& b) { return CReal(a) + b; } \
  inline CReal operator-(const CReal& a, T b) { return a - CReal(b); } inline CReal operator-(T a, const CReal& b) { return CReal(a) - b; } \
  inline CReal operator*(const CReal& a, T b) { return a * CReal(b); } inline CReal operator*(T a, const CReal& b) { return CReal(a) * b; } \
  inline CReal operator/(const CReal& a, T b) { return a / CReal(b); } inline CReal operator/(T a, const CReal& b) { return CReal(a) / b; } \
  inline bool operator<(const CReal& a, T b) { return a < CReal(b); } inline bool operator<(T a, const CReal& b) { return CReal(a) < b; }   \
  inline bool operator>(const CReal& a, T b) { return a > CReal(b); } inline bool operator>(T a, const CReal& b) { return CReal(a) > b; }   \
  inline bool operator<=(const CReal& a, T b) { return a <= CReal(b); } inline bool operator<=(T a, const CReal& b) { return CReal(a) <= b; } \
  inline bool operator>=(const CReal& a, T b) { return a >= CReal(b); } inline bool operator>=(T a, const CReal& b) { return CReal(a) >= b; } \
  inline bool operator==(const CReal& a, T b) { return a == CReal(b); } inline bool operator!=(const CReal& a, T b) { return a != CReal(b); }
CR_MIXED(double)
CR_MIXED(float)
CR_MIXED(int)
