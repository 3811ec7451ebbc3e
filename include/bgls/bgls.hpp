// C++ host-side mirror of the reference's Go package `bgls` on the hot path (bgls/bgls.go,
// bgls/blsKosk.go): same function names and semantics, each verification = ONE batch C call.
#pragma once
#include "curves.hpp"

namespace bgls_go {   // "package bgls"; the name bgls:: is taken by the kernels' namespace

using curves::Bytes;
using curves::CurveSystem;
using curves::Point;

// bgls/bgls.go:40-43
inline Point LoadPublicKey(const CurveSystem* curve, const Bytes& sk_be32) { return curve->GetG2().Mul(sk_be32); }
// bgls/bgls.go:46-56
inline Point Sign(const CurveSystem* curve, const Bytes& sk_be32, const Bytes& msg) { return curve->HashToG1(msg).Mul(sk_be32); }
// bgls/blsKosk.go:73-77
inline Point KoskSign(const CurveSystem* curve, const Bytes& sk_be32, const Bytes& msg) {
  Bytes m(1, 1);
  m.insert(m.end(), msg.begin(), msg.end());
  return Sign(curve, sk_be32, m);
}
// bgls/bgls.go:123-131
inline Point AggregateSignatures(const std::vector<Point>& sigs) { return curves::AggregatePoints(sigs); }
inline Point AggregateKeys(const std::vector<Point>& keys) { return curves::AggregatePoints(keys); }

// verifyAggSig, bgls/bgls.go:94-119
inline bool verifyAggSig(const CurveSystem* curve, const Point& aggsig, const std::vector<Point>& keys,
                         const std::vector<Bytes>& msgs, bool allowDuplicates) {
  if (keys.size() != msgs.size()) return false;
  if (aggsig.curve != curve || aggsig.group != BGLS_G1) return false;
  Bytes kb, blob;
  std::vector<uint64_t> off(msgs.size() + 1, 0);
  for (size_t i = 0; i < keys.size(); ++i) {
    if (keys[i].curve != curve || keys[i].group != BGLS_G2) return false;
    kb.insert(kb.end(), keys[i].raw.begin(), keys[i].raw.end());
    off[i] = blob.size();
    blob.insert(blob.end(), msgs[i].begin(), msgs[i].end());
  }
  off[msgs.size()] = blob.size();
  return bgls_verify_aggregate(curve->id, aggsig.raw.data(), kb.data(), blob.data(), off.data(), keys.size(), allowDuplicates ? 1 : 0) == 1;
}
// bgls/bgls.go:82-84
inline bool VerifyAggregateSignature(const CurveSystem* curve, const Point& aggsig, const std::vector<Point>& keys,
                                     const std::vector<Bytes>& msgs) {
  return verifyAggSig(curve, aggsig, keys, msgs, false);
}
// bgls/blsKosk.go:100-106
inline bool KoskVerifyAggregateSignature(const CurveSystem* curve, const Point& aggsig, const std::vector<Point>& keys,
                                         const std::vector<Bytes>& msgs) {
  std::vector<Bytes> m2;
  for (const Bytes& m : msgs) {
    Bytes x(1, 1);
    x.insert(x.end(), m.begin(), m.end());
    m2.push_back(x);
  }
  return verifyAggSig(curve, aggsig, keys, m2, true);
}
// verifyMultiSignature, bgls/bgls.go:89-92
inline bool verifyMultiSignature(const CurveSystem* curve, const Point& aggsig, const std::vector<Point>& keys, const Bytes& msg) {
  if (aggsig.curve != curve || aggsig.group != BGLS_G1) return false;      // a nil / foreign aggsig is `false`
  Bytes kb;
  for (const Point& k : keys) {
    if (k.curve != curve || k.group != BGLS_G2) return false;
    kb.insert(kb.end(), k.raw.begin(), k.raw.end());
  }
  return bgls_verify_multi(curve->id, aggsig.raw.data(), kb.data(), keys.size(), msg.data(), msg.size()) == 1;
}
// bgls/bgls.go:59-70
inline bool VerifySingleSignature(const CurveSystem* curve, const Point& sig, const Point& pubKey, const Bytes& msg) {
  return verifyMultiSignature(curve, sig, {pubKey}, msg);
}
// bgls/blsKosk.go:117-120
inline bool KoskVerifyMultiSignature(const CurveSystem* curve, const Point& aggsig, const std::vector<Point>& keys, const Bytes& msg) {
  Bytes m(1, 1);
  m.insert(m.end(), msg.begin(), msg.end());
  return verifyMultiSignature(curve, aggsig, keys, m);
}

// A []Point of public keys resident on the GPU(s) (bgls_keys_t): uploaded, parsed and validated once, verified many times.
class KeySet {
 public:
  KeySet(const CurveSystem* curve, const std::vector<Point>& keys, const std::vector<int>& devices = {}, bool check = true) : curve_(curve), n_(keys.size()) {
    Bytes kb;
    for (const Point& k : keys) {
      if (k.curve != curve || k.group != BGLS_G2) return;
      kb.insert(kb.end(), k.raw.begin(), k.raw.end());
    }
    const int nd = devices.empty() ? 1 : (int)devices.size();
    ok_ = bgls_keys_upload(curve->id, kb.data(), n_, devices.empty() ? nullptr : devices.data(), nd, check ? BGLS_KEYS_CHECK : 0u, &h_) == 0;
  }
  ~KeySet() { if (ok_) bgls_keys_free(h_); }
  KeySet(const KeySet&) = delete;
  KeySet& operator=(const KeySet&) = delete;
  bool ok() const { return ok_; }
  // verifyAggSig (bgls/bgls.go:94-119) / verifyMultiSignature (bgls/bgls.go:89-92) against the resident keys
  bool VerifyAggregateSignature(const Point& aggsig, const std::vector<Bytes>& msgs, bool allowDuplicates = false) const {
    if (!ok_ || msgs.size() != n_ || aggsig.curve != curve_ || aggsig.group != BGLS_G1) return false;
    Bytes blob;
    std::vector<uint64_t> off(msgs.size() + 1, 0);
    for (size_t i = 0; i < msgs.size(); ++i) {
      blob.insert(blob.end(), msgs[i].begin(), msgs[i].end());
      off[i + 1] = blob.size();
    }
    return bgls_verify_aggregate_h(h_, aggsig.raw.data(), blob.data(), off.data(), n_, allowDuplicates ? 1 : 0) == 1;
  }
  bool VerifyMultiSignature(const Point& aggsig, const Bytes& msg) const {
    if (!ok_ || aggsig.curve != curve_ || aggsig.group != BGLS_G1) return false;
    return bgls_verify_multi_h(h_, aggsig.raw.data(), msg.data(), msg.size()) == 1;
  }

 private:
  const CurveSystem* curve_;
  size_t n_;
  bgls_keys_t h_ = 0;
  bool ok_ = false;
};

// ---- hashed aggregation exponents (bgls/blsHAE.go) and multiplicities (bgls/blsKosk.go:137-150) ----
namespace detail {
inline bool g2_bytes(const CurveSystem* curve, const std::vector<Point>& keys, Bytes& kb) {
  for (const Point& k : keys) {
    if (k.curve != curve || k.group != BGLS_G2) return false;
    kb.insert(kb.end(), k.raw.begin(), k.raw.end());
  }
  return true;
}
}  // namespace detail
// hashPubKeysToExponents, bgls/blsHAE.go:80-93: n exponents as 16-byte big-endian strings
inline std::vector<Bytes> hashPubKeysToExponents(const std::vector<Point>& pubkeys) {
  std::vector<Bytes> t;
  if (pubkeys.empty()) return t;
  Bytes kb, out(16 * pubkeys.size());
  if (!detail::g2_bytes(pubkeys[0].curve, pubkeys, kb)) return t;
  if (bgls_hae_exponents(pubkeys[0].curve->id, kb.data(), pubkeys.size(), out.data()) != 0) return t;
  for (size_t i = 0; i < pubkeys.size(); ++i) t.emplace_back(out.begin() + 16 * i, out.begin() + 16 * (i + 1));
  return t;
}
// AggregateSignaturesWithHAE, bgls/blsHAE.go:39-46 (invalid Point = nil on a length mismatch)
inline Point AggregateSignaturesWithHAE(const std::vector<Point>& sigs, const std::vector<Point>& pubkeys) {
  if (sigs.size() != pubkeys.size() || sigs.empty()) return Point{};
  const CurveSystem* curve = sigs[0].curve;
  Bytes sb, kb;
  for (const Point& s : sigs) {
    if (s.curve != curve || s.group != BGLS_G1) return Point{};
    sb.insert(sb.end(), s.raw.begin(), s.raw.end());
  }
  if (!detail::g2_bytes(curve, pubkeys, kb)) return Point{};
  Bytes out(curve->size(BGLS_G1));
  if (bgls_aggregate_signatures_hae(curve->id, sb.data(), kb.data(), sigs.size(), out.data()) != 0) return Point{};
  return Point{curve, BGLS_G1, out};
}
// VerifyAggregateSignatureWithHAE, bgls/blsHAE.go:49-53
inline bool VerifyAggregateSignatureWithHAE(const CurveSystem* curve, const Point& aggsig, const std::vector<Point>& pubkeys,
                                            const std::vector<Bytes>& msgs) {
  if (pubkeys.size() != msgs.size() || aggsig.curve != curve || aggsig.group != BGLS_G1) return false;
  Bytes kb, blob;
  if (!detail::g2_bytes(curve, pubkeys, kb)) return false;
  std::vector<uint64_t> off(msgs.size() + 1, 0);
  for (size_t i = 0; i < msgs.size(); ++i) {
    blob.insert(blob.end(), msgs[i].begin(), msgs[i].end());
    off[i + 1] = blob.size();
  }
  return bgls_verify_aggregate_hae(curve->id, aggsig.raw.data(), kb.data(), blob.data(), off.data(), msgs.size()) == 1;
}
// VerifyMultiSignatureWithHAE, bgls/blsHAE.go:56-58
inline bool VerifyMultiSignatureWithHAE(const CurveSystem* curve, const Point& aggsig, const std::vector<Point>& pubkeys, const Bytes& msg) {
  Bytes kb;
  if (aggsig.curve != curve || aggsig.group != BGLS_G1 || !detail::g2_bytes(curve, pubkeys, kb)) return false;
  return bgls_verify_multi_hae(curve->id, aggsig.raw.data(), kb.data(), pubkeys.size(), msg.data(), msg.size()) == 1;
}
// KoskVerifyBatchMultiSignature, bgls/blsKosk.go:126-133: one call -- every key set summed in one launch, ONE aggregate verification
inline bool KoskVerifyBatchMultiSignature(const CurveSystem* curve, const std::vector<Point>& aggsigs, const std::vector<std::vector<Point>>& pubkeys,
                                          const std::vector<Bytes>& msgs) {
  if (aggsigs.size() != pubkeys.size() || pubkeys.size() != msgs.size() || msgs.empty()) return false;
  Bytes sb, kb, blob;
  std::vector<uint64_t> koff(pubkeys.size() + 1, 0), moff(msgs.size() + 1, 0);
  for (size_t i = 0; i < msgs.size(); ++i) {
    if (aggsigs[i].curve != curve || aggsigs[i].group != BGLS_G1) return false;
    sb.insert(sb.end(), aggsigs[i].raw.begin(), aggsigs[i].raw.end());
    Bytes one;
    if (!detail::g2_bytes(curve, pubkeys[i], one)) return false;
    kb.insert(kb.end(), one.begin(), one.end());
    koff[i + 1] = koff[i] + pubkeys[i].size();
    blob.push_back(1);                                   // the Kosk prefix (blsKosk.go:100-106)
    blob.insert(blob.end(), msgs[i].begin(), msgs[i].end());
    moff[i + 1] = blob.size();
  }
  return bgls_verify_multi_batch(curve->id, sb.data(), kb.data(), koff.data(), msgs.size(), blob.data(), moff.data(), 1) == 1;
}
// KoskVerifyMultiSignatureWithMultiplicity, bgls/blsKosk.go:137-150 (multiplicity == nullptr: plain KoskVerifyMultiSignature)
inline bool KoskVerifyMultiSignatureWithMultiplicity(const CurveSystem* curve, const Point& aggsig, const std::vector<Point>& keys,
                                                     const std::vector<int64_t>* multiplicity, const Bytes& msg) {
  if (!multiplicity) return KoskVerifyMultiSignature(curve, aggsig, keys, msg);
  if (keys.size() != multiplicity->size()) return false;
  Bytes kb, m(1, 1);
  if (aggsig.curve != curve || aggsig.group != BGLS_G1 || !detail::g2_bytes(curve, keys, kb)) return false;
  m.insert(m.end(), msg.begin(), msg.end());
  return bgls_verify_multi_multiplicity(curve->id, aggsig.raw.data(), kb.data(), multiplicity->data(), keys.size(), m.data(), m.size()) == 1;
}

}  // namespace bgls_go
