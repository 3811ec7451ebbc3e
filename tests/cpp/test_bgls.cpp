// The reference's scheme tests through the C++ host mirror (include/bgls/*.hpp == Go packages
// `curves` / `bgls` on this path): bgls/bgls_test.go:19-77 (TestSingleSigner, TestAggregation) and
// bgls/blsKosk_test.go:35-64 (TestKoskMultiSig).  Built and run by tests/test_gpu_cpp_mirror.py.
#include <cstdio>
#include <random>
#include "bgls/bgls.hpp"

using namespace curves;
using namespace bgls_go;

static std::mt19937_64 rng(20260928);
static Bytes randBytes(size_t n) { Bytes b(n); for (auto& x : b) x = (uint8_t)rng(); return b; }
static Bytes randScalar() { Bytes b = randBytes(32); b[0] &= 0x0f; return b; }   // < 2^252 < order
static int failures = 0;
#define CHECK(cond, what) do { if (!(cond)) { std::printf("FAIL %s: %s\n", curve->Name().c_str(), what); ++failures; } } while (0)

static void TestSingleSigner(const CurveSystem* curve) {
  Bytes sk = randScalar(); Point vk = LoadPublicKey(curve, sk);
  Bytes d = randBytes(64);
  Point sig = Sign(curve, sk, d);
  CHECK(VerifySingleSignature(curve, sig, vk, d), "Standard BLS signature verification failed");
  auto sig2 = sig.Copy().Add(curve->GetG1());
  CHECK(sig2.second && !VerifySingleSignature(curve, sig2.first, vk, d), "verification succeeding when it shouldn't");
}

static void TestAggregation(const CurveSystem* curve) {
  const int N = 6, Size = 32;
  std::vector<Bytes> msgs; std::vector<Point> sigs, pubkeys;
  for (int i = 0; i < N; ++i) {
    msgs.push_back(randBytes(Size));
    Bytes sk = randScalar();
    pubkeys.push_back(LoadPublicKey(curve, sk)); sigs.push_back(Sign(curve, sk, msgs[i]));
  }
  Point aggSig = AggregateSignatures(sigs);
  CHECK(VerifyAggregateSignature(curve, aggSig, pubkeys, msgs), "Aggregate Point1 verification failed");
  std::vector<Point> fewer(pubkeys.begin(), pubkeys.end() - 1);
  CHECK(!VerifyAggregateSignature(curve, aggSig, fewer, msgs), "succeeding without enough pubkeys");
  Bytes skf = randScalar();
  std::vector<Point> pk2 = pubkeys, sg2 = sigs; std::vector<Bytes> m2 = msgs;
  pk2.push_back(LoadPublicKey(curve, skf)); sg2.push_back(Sign(curve, skf, msgs[0])); m2.push_back(msgs[0]);
  Point agg2 = AggregateSignatures(sg2);
  CHECK(!VerifyAggregateSignature(curve, agg2, pk2, m2), "succeeding with duplicate messages");
  CHECK(KoskVerifyAggregateSignature(curve, AggregateSignatures({KoskSign(curve, skf, msgs[0])}), {pk2.back()}, {msgs[0]}), "Kosk aggregate failed");
  CHECK(!VerifyAggregateSignature(curve, agg2, pubkeys, msgs), "succeeding with invalid signature");
  std::vector<Bytes> sw = msgs; sw[0] = msgs[1]; sw[1] = msgs[0];
  CHECK(!VerifyAggregateSignature(curve, aggSig, pubkeys, sw), "succeeded with messages 0 and 1 switched");
}

static void TestKoskMultiSig(const CurveSystem* curve) {
  const int Signers = 8;
  Bytes msg = randBytes(32);
  std::vector<Point> signers, sigs;
  for (int j = 0; j < Signers; ++j) { Bytes sk = randScalar(); sigs.push_back(KoskSign(curve, sk, msg)); signers.push_back(LoadPublicKey(curve, sk)); }
  Point aggsig = AggregateSignatures(sigs);
  CHECK(KoskVerifyMultiSignature(curve, aggsig, signers, msg), "Aggregate MultiSig verification failed");
  CHECK(!KoskVerifyMultiSignature(curve, aggsig, signers, randBytes(32)), "succeeded on incorrect msg");
  Point aggkey = AggregateKeys(signers);
  CHECK(KoskVerifyMultiSignature(curve, aggsig, {aggkey}, msg), "failed with the aggkey pre-aggregated");
  signers[0] = LoadPublicKey(curve, randScalar());
  CHECK(!KoskVerifyMultiSignature(curve, aggsig, signers, msg), "succeeded on incorrect signers");
  auto bad = curve->PairingProduct({curve->GetG1()}, {curve->GetG2(), curve->GetG2()});
  CHECK(!bad.second, "PairingProduct length mismatch must be (nil,false)");
  auto idt = curve->Pair(curve->GetG1(), curve->GetG2Infinity());
  CHECK(idt.second && idt.first.Equals(curve->GetGTIdentity()), "Pair(g1, inf) must be the GT identity");
}

// bgls/blsHAE_test.go:58-82 TestMultiSigWithHAE + :14-56 TestAggregationWithHAE (one trial each) and the multiplicity path
static void TestHAE(const CurveSystem* curve) {
  const int Signers = 6;
  Bytes msg = randBytes(32);
  std::vector<Point> signers, sigs;
  for (int j = 0; j < Signers; ++j) { Bytes sk = randScalar(); sigs.push_back(Sign(curve, sk, msg)); signers.push_back(LoadPublicKey(curve, sk)); }
  CHECK(hashPubKeysToExponents(signers).size() == (size_t)Signers, "exponent count");
  Point aggSig = AggregateSignaturesWithHAE(sigs, signers);
  CHECK(aggSig.valid() && VerifyMultiSignatureWithHAE(curve, aggSig, signers, msg), "HAE MultiSig verification failed");
  CHECK(!VerifyMultiSignatureWithHAE(curve, aggSig, signers, randBytes(32)), "HAE MultiSig succeeded on incorrect msg");
  CHECK(!VerifyMultiSignatureWithHAE(curve, AggregateSignatures(sigs), signers, msg), "plain aggregate accepted by the HAE verifier");
  std::vector<Point> fewer(signers.begin(), signers.end() - 1);
  CHECK(!AggregateSignaturesWithHAE(sigs, fewer).valid(), "aggregation succeeded with differing numbers of signatures and pubkeys");
  std::vector<Bytes> msgs; std::vector<Point> s2;
  for (int j = 0; j < Signers; ++j) { msgs.push_back(j == Signers - 1 ? msgs[0] : randBytes(32)); }
  std::vector<Bytes> sks; std::vector<Point> pk2;
  for (int j = 0; j < Signers; ++j) { Bytes sk = randScalar(); s2.push_back(Sign(curve, sk, msgs[j])); pk2.push_back(LoadPublicKey(curve, sk)); }
  Point agg2 = AggregateSignaturesWithHAE(s2, pk2);
  CHECK(VerifyAggregateSignatureWithHAE(curve, agg2, pk2, msgs), "HAE aggregate failing with duplicate messages");
  std::vector<Bytes> sw = msgs; sw[1] = msgs[2]; sw[2] = msgs[1];
  CHECK(!VerifyAggregateSignatureWithHAE(curve, agg2, pk2, sw), "HAE aggregate succeeded with messages switched");
  // multiplicities
  std::vector<int64_t> mult = {2, 1, -1, 0, 3, 1};
  std::vector<Point> ks, scaled;
  Bytes kmsg = randBytes(32);
  std::vector<Point> ksign;
  for (int j = 0; j < Signers; ++j) { Bytes sk = randScalar(); ks.push_back(LoadPublicKey(curve, sk)); scaled.push_back(KoskSign(curve, sk, kmsg).MulInt(mult[j])); }
  Point am = AggregateSignatures(scaled);
  CHECK(KoskVerifyMultiSignatureWithMultiplicity(curve, am, ks, &mult, kmsg), "multiplicity verification failed");
  mult[0] += 1;
  CHECK(!KoskVerifyMultiSignatureWithMultiplicity(curve, am, ks, &mult, kmsg), "multiplicity verification succeeded on wrong factors");
}

// resident key sets (bgls_keys_t) through the C++ mirror: same verdicts as the []Point calls, on 1 and 3 shards of device 0;
// GT exponentiation (PointT.Mul): e(g1, g2)^k == e(k g1, g2) (the bilinearity check of curves/curve_test.go:120-141)
static void TestKeySetAndGtMul(const CurveSystem* curve) {
  const int N = 17;
  std::vector<Point> keys, sigs;
  std::vector<Bytes> msgs;
  for (int j = 0; j < N; ++j) { Bytes sk = randScalar(); msgs.push_back(randBytes(24 + j)); sigs.push_back(Sign(curve, sk, msgs.back())); keys.push_back(LoadPublicKey(curve, sk)); }
  Point agg = AggregateSignatures(sigs);
  CHECK(VerifyAggregateSignature(curve, agg, keys, msgs), "aggregate verification failed");
  for (int shards : {1, 3}) {
    KeySet ks(curve, keys, std::vector<int>(shards, 0));
    CHECK(ks.ok(), "key-set upload failed");
    CHECK(ks.VerifyAggregateSignature(agg, msgs), "key-set aggregate verification failed");
    std::vector<Bytes> sw = msgs; std::swap(sw[0], sw[N - 1]);
    CHECK(!ks.VerifyAggregateSignature(agg, sw), "key-set aggregate verification succeeded with messages switched");
    std::vector<Bytes> fewer(msgs.begin(), msgs.end() - 1);
    CHECK(!ks.VerifyAggregateSignature(agg, fewer), "key-set aggregate verification succeeded with a missing message");
  }
  Bytes m = randBytes(32);
  std::vector<Point> ksigs;
  std::vector<Point> kkeys;
  for (int j = 0; j < N; ++j) { Bytes sk = randScalar(); ksigs.push_back(KoskSign(curve, sk, m)); kkeys.push_back(LoadPublicKey(curve, sk)); }
  Bytes km(1, 1); km.insert(km.end(), m.begin(), m.end());
  KeySet kk(curve, kkeys, {0, 0});
  CHECK(kk.VerifyMultiSignature(AggregateSignatures(ksigs), km), "key-set multi-signature verification failed");
  CHECK(!kk.VerifyMultiSignature(AggregateSignatures(ksigs), m), "key-set multi-signature verification succeeded on the wrong message");
  Bytes k = randScalar();
  auto e = curve->Pair(curve->GetG1(), curve->GetG2());
  auto ek = curve->Pair(curve->GetG1().Mul(k), curve->GetG2());
  CHECK(e.second && ek.second && e.first.Mul(k).Equals(ek.first), "e(g1, g2)^k != e(k g1, g2)");
  CHECK(e.first.Mul(k, true).Add(ek.first).first.Equals(curve->GetGTIdentity()), "e(g1, g2)^-k * e(k g1, g2) != 1");
}

// curves/curve_test.go:73-84 TestMarshal: Unmarshal(Marshal(P)) == P and Unmarshal(MarshalUncompressed(P)) == P, G1 and G2,
// both curves (alt-bn128: the reference's own compressed format; BLS12-381: 48 / 96 bytes, ebfull/pairing layout)
static void TestMarshal(const CurveSystem* curve) {
  for (int j = 0; j < 4; ++j) {
    Bytes k = randScalar();
    Point p1 = curve->GetG1().Mul(k), p2 = curve->GetG2().Mul(k);
    Bytes m1 = p1.Marshal(), m2 = p2.Marshal();
    CHECK(2 * m1.size() == p1.raw.size() && 2 * m2.size() == p2.raw.size(), "Marshal: wrong length");
    auto u1 = curve->UnmarshalG1(m1), u2 = curve->UnmarshalG2(m2);
    CHECK(u1.second && u1.first.Equals(p1), "UnmarshalG1(Marshal(P)) != P");
    CHECK(u2.second && u2.first.Equals(p2), "UnmarshalG2(Marshal(P)) != P");
    auto w1 = curve->UnmarshalG1(p1.MarshalUncompressed()), w2 = curve->UnmarshalG2(p2.MarshalUncompressed());
    CHECK(w1.second && w1.first.Equals(p1) && w2.second && w2.first.Equals(p2), "Unmarshal(MarshalUncompressed(P)) != P");
  }
  CHECK(!curve->UnmarshalG1(Bytes(7, 1)).second, "UnmarshalG1 accepted 7 bytes");
}

int main() {
  if (bgls_init(0) != 0) { std::printf("bgls_init: %s\n", bgls_last_error()); return 2; }
  for (const CurveSystem* curve : {Altbn128(), Bls12()}) { TestSingleSigner(curve); TestAggregation(curve); TestKoskMultiSig(curve); TestHAE(curve); TestKeySetAndGtMul(curve); TestMarshal(curve); }
  std::printf(failures ? "FAILED %d\n" : "ALL OK\n", failures);
  return failures ? 1 : 0;
}
