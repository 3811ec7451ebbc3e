// C++ host-side mirror of the reference's Go package `curves` on the hot path
// (curves/curve.go:12-70: CurveSystem / Point / PointT, plus AggregatePoints :73-121 and
// ScalePoints :190-214), header-only over the C ABI (include/bgls_hip.h).  Same names, argument
// meaning and error behaviour: fallible operations return (value, ok) pairs, never throw for bad data.
// All arithmetic happens in the HIP library; nothing here computes field or group operations.
#pragma once
#include <cstdint>
#include <cstring>
#include <memory>
#include <string>
#include <utility>
#include <vector>

#include "../bgls_hip.h"

namespace curves {

typedef std::vector<uint8_t> Bytes;
struct CurveSystem;

// curves.Point (curves/curve.go:51-59), uncompressed wire bytes
struct Point {
  const CurveSystem* curve = nullptr;
  int group = 0;  // BGLS_G1 / BGLS_G2
  Bytes raw;
  bool valid() const { return curve != nullptr; }
  std::pair<Point, bool> Add(const Point& o) const;
  Point Copy() const { return *this; }
  bool Equals(const Point& o) const { return curve == o.curve && group == o.group && raw == o.raw; }
  Bytes MarshalUncompressed() const { return raw; }
  // Point.Marshal: the compressed form (curves/altbn128.go:81-89,203-221; curves/bls12_381.go:54-62,115-123 -- BLS12-381 in the
  // ebfull/pairing layout, see bgls_compress_points); empty on failure
  Bytes Marshal() const;
  // scalar: 32-byte big-endian magnitude; negative => negate-then-multiply (curves/altbn128.go:107-121)
  Point Mul(const Bytes& magnitude_be32, bool negative = false) const;
  Point MulInt(long long k) const;
};

// curves.PointT (curves/curve.go:62-70)
struct PointT {
  const CurveSystem* curve = nullptr;
  Bytes raw;
  bool valid() const { return curve != nullptr; }
  std::pair<PointT, bool> Add(const PointT& o) const;
  PointT Copy() const { return *this; }
  bool Equals(const PointT& o) const { return curve == o.curve && raw == o.raw; }
  Bytes Marshal() const { return raw; }
  // PointT.Mul (curves/altbn128.go:273-281): exponentiation in GT; an invalid PointT on failure, never an exception
  PointT Mul(const Bytes& magnitude_be32, bool negative = false) const;
};

// curves.CurveSystem (curves/curve.go:12-49)
struct CurveSystem {
  int id;
  std::string name;
  std::string Name() const { return name; }
  size_t size(int group) const { return group == BGLS_G1 ? bgls_g1_size(id) : bgls_g2_size(id); }

  std::pair<Point, bool> Unmarshal(int group, const Bytes& d) const {
    if (2 * d.size() == size(group)) {      // compressed branch (curves/altbn128.go:296-376, curves/bls12_381.go:242-264)
      Bytes out(size(group));
      uint8_t ok = 0;
      if (bgls_decompress_points(id, group, d.data(), 1, out.data(), &ok) != 0 || ok != 1) return {Point{}, false};
      return {Point{this, group, out}, true};
    }
    if (d.size() != size(group) || bgls_point_check(id, group, d.data()) != 1) return {Point{}, false};
    return {Point{this, group, d}, true};
  }
  std::pair<Point, bool> UnmarshalG1(const Bytes& d) const { return Unmarshal(BGLS_G1, d); }
  std::pair<Point, bool> UnmarshalG2(const Bytes& d) const { return Unmarshal(BGLS_G2, d); }
  std::pair<PointT, bool> UnmarshalGT(const Bytes& d) const {
    if (d.size() != bgls_gt_size(id)) return {PointT{}, false};
    return {PointT{this, d}, true};
  }
  Point Generator(int group) const {
    Bytes o(size(group));
    bgls_generator(id, group, o.data());
    return Point{this, group, o};
  }
  Point GetG1() const { return Generator(BGLS_G1); }
  Point GetG2() const { return Generator(BGLS_G2); }
  Point GetG1Infinity() const { return Point{this, BGLS_G1, Bytes(size(BGLS_G1), 0)}; }
  Point GetG2Infinity() const { return Point{this, BGLS_G2, Bytes(size(BGLS_G2), 0)}; }
  PointT GetGTIdentity() const {
    Bytes o(bgls_gt_size(id));
    bgls_gt_identity(id, o.data());
    return PointT{this, o};
  }
  Point HashToG1(const Bytes& message) const {
    uint64_t off[2] = {0, message.size()};
    Bytes o(size(BGLS_G1));
    if (bgls_hash_to_g1(id, message.data(), off, 1, o.data()) != 0) return Point{};
    return Point{this, BGLS_G1, o};
  }
  std::pair<PointT, bool> Pair(const Point& a, const Point& b) const { return PairingProduct({a}, {b}); }
  // one C call for the whole slice (replaces concurrentPairingProduct, curves/curve.go:125-170)
  std::pair<PointT, bool> PairingProduct(const std::vector<Point>& p1, const std::vector<Point>& p2) const {
    if (p1.size() != p2.size()) return {PointT{}, false};
    Bytes g1, g2;
    for (size_t i = 0; i < p1.size(); ++i) {
      if (p1[i].curve != this || p2[i].curve != this || p1[i].group != BGLS_G1 || p2[i].group != BGLS_G2) return {PointT{}, false};
      g1.insert(g1.end(), p1[i].raw.begin(), p1[i].raw.end());
      g2.insert(g2.end(), p2[i].raw.begin(), p2[i].raw.end());
    }
    Bytes o(bgls_gt_size(id));
    if (bgls_pairing_product(id, g1.data(), g2.data(), p1.size(), o.data()) != 0) return {PointT{}, false};
    return {PointT{this, o}, true};
  }
};

inline const CurveSystem* Altbn128() {
  static const CurveSystem c{BGLS_CURVE_ALTBN128, "altbn128"};
  return &c;
}
inline const CurveSystem* Bls12() {
  static const CurveSystem c{BGLS_CURVE_BLS12_381, "bls12"};
  return &c;
}

inline std::pair<Point, bool> Point::Add(const Point& o) const {
  if (curve != o.curve || group != o.group) return {Point{}, false};
  Bytes out(raw.size());
  if (bgls_point_add(curve->id, group, raw.data(), o.raw.data(), out.data()) != 0) return {Point{}, false};
  return {Point{curve, group, out}, true};
}
inline PointT PointT::Mul(const Bytes& mag, bool negative) const {
  if (!curve || mag.size() != 32) return PointT{};
  Bytes o(raw.size());
  if (bgls_gt_pow(curve->id, raw.data(), mag.data(), negative ? 1 : 0, o.data()) != 0) return PointT{};
  return PointT{curve, o};
}
inline Bytes Point::Marshal() const {
  if (!valid()) return Bytes();
  Bytes out(raw.size() / 2);
  if (bgls_compress_points(curve->id, group, raw.data(), 1, out.data()) != 0) return Bytes();
  return out;
}
inline Point Point::Mul(const Bytes& mag, bool negative) const {
  uint8_t sign = negative ? 1 : 0;
  Bytes out(raw.size());
  if (mag.size() != 32 || bgls_scale_points(curve->id, group, raw.data(), mag.data(), &sign, 1, out.data()) != 0) return Point{};
  return Point{curve, group, out};
}
inline Point Point::MulInt(long long k) const {
  Bytes m(32, 0);
  unsigned long long a = k < 0 ? 0ull - (unsigned long long)k : (unsigned long long)k;
  for (int i = 0; i < 8; ++i) m[31 - i] = (uint8_t)(a >> (8 * i));
  return Mul(m, k < 0);
}
inline std::pair<PointT, bool> PointT::Add(const PointT& o) const {
  if (curve != o.curve) return {PointT{}, false};
  Bytes out(raw.size());
  if (bgls_gt_mul(curve->id, raw.data(), o.raw.data(), out.data()) != 0) return {PointT{}, false};
  return {PointT{curve, out}, true};
}

// curves.AggregatePoints (curves/curve.go:73-121) as one device call; empty input yields an invalid Point
inline Point AggregatePoints(const std::vector<Point>& pts) {
  if (pts.empty()) return Point{};
  Bytes in;
  for (const Point& p : pts) in.insert(in.end(), p.raw.begin(), p.raw.end());
  Bytes out(pts[0].raw.size());
  if (bgls_aggregate_points(pts[0].curve->id, pts[0].group, in.data(), pts.size(), out.data()) != 0) return Point{};
  return Point{pts[0].curve, pts[0].group, out};
}

}  // namespace curves
