// Copyright (C) 2018 Authors
// distributed under Apache 2.0 license
//
// curves_hip.go -- cgo binding of the MI355X engine (include/bgls_hip.h) behind the reference's
// CurveSystem / Point / PointT interfaces (curves/curve.go:12-70).  Lives INSIDE package curves
// because CurveSystem has unexported methods (curves/curve.go:38-44).
//
// UNVERIFIED TEXT: THIS FILE HAS NEVER MET A GO COMPILER (no Go toolchain in the build container, no network for the
// reference's un-vendored modules).  It is the binding a maintainer would start from, written against the C ABI that IS
// tested (ctypes in tests/, the C++ mirror in tests/cpp) -- not a tested drop-in.  The ABI never lets a C++ exception
// unwind into a cgo frame (every entry point is guarded; BGLS_ERR_NOMEM = -6 since round 5; anything but 1 maps to false).  Build:  CGO_CFLAGS=-I$REPO/include CGO_LDFLAGS="-L$REPO/bgls_amd -lbgls_hip" go build ./...
package curves

/*
#cgo LDFLAGS: -lbgls_hip
#include <stdlib.h>
#include "bgls_hip.h"
*/
import "C"

import (
	"bytes"
	"math/big"
	"runtime"
	"unsafe"
)

type hipCurve struct {
	id   C.int
	name string
	base CurveSystem // the pure-Go curve: constants and the big.Int helpers only, never pairings
}

// AltbnHip / Bls12Hip are drop-in replacements for curves.Altbn128 / curves.Bls12.
var AltbnHip = &hipCurve{C.BGLS_CURVE_ALTBN128, "altbn128", Altbn128}
var Bls12Hip = &hipCurve{C.BGLS_CURVE_BLS12_381, "bls12", Bls12}

type hipPoint struct {
	c     *hipCurve
	group C.int  // C.BGLS_G1 or C.BGLS_G2
	raw   []byte // uncompressed wire bytes, the reference's own format
}
type hipPointT struct {
	c   *hipCurve
	raw []byte
}

func p(b []byte) *C.uint8_t {
	if len(b) == 0 {
		return nil
	}
	return (*C.uint8_t)(unsafe.Pointer(&b[0]))
}

func (c *hipCurve) size(group C.int) int {
	if group == C.BGLS_G1 {
		return int(C.bgls_g1_size(c.id))
	}
	return int(C.bgls_g2_size(c.id))
}

// ---- Point (curves/curve.go:51-59) -------------------------------------------------------
func (pt *hipPoint) Add(o Point) (Point, bool) {
	q, ok := o.(*hipPoint)
	if !ok || q.c != pt.c || q.group != pt.group {
		return nil, false
	}
	out := make([]byte, len(pt.raw))
	if C.bgls_point_add(pt.c.id, pt.group, p(pt.raw), p(q.raw), p(out)) != 0 {
		return nil, false
	}
	return &hipPoint{pt.c, pt.group, out}, true
}
func (pt *hipPoint) Copy() Point { return &hipPoint{pt.c, pt.group, append([]byte(nil), pt.raw...)} }
func (pt *hipPoint) Equals(o Point) bool {
	q, ok := o.(*hipPoint)
	return ok && q.c == pt.c && q.group == pt.group && bytes.Equal(q.raw, pt.raw)
}
func (pt *hipPoint) MarshalUncompressed() []byte { return append([]byte(nil), pt.raw...) }

// Marshal: the compressed form.  alt-bn128: the reference's own 32 / 64-byte format (curves/altbn128.go:81-89,203-221).
// BLS12-381: 48 / 96 bytes in the ebfull/pairing layout the reference names as its target (curves/bls12_381.go:54-62,115-123
// "TODO Make this match ebfull/pairing marshalling"); byte parity with the un-vendored dis2/bls12 is unpinned (bgls_hip.h).
func (pt *hipPoint) Marshal() []byte {
	out := make([]byte, len(pt.raw)/2)
	if C.bgls_compress_points(pt.c.id, pt.group, p(pt.raw), 1, p(out)) != 0 {
		return nil
	}
	return out
}
// scalar32 returns |k| as the 32-byte big-endian magnitude the C ABI takes.  Magnitudes of 2^256 or more are reduced
// modulo the group order first (on points of order r -- every validated Point, every GT element -- that is the same
// multiple); the caller's big.Int is never mutated (unlike curves/bls12_381.go:70) and FillBytes never panics.
func (c *hipCurve) scalar32(k *big.Int) ([]byte, C.int) {
	neg := C.int(0)
	mag := new(big.Int).Abs(k)
	if k.Sign() < 0 {
		neg = 1
	}
	if mag.BitLen() > 256 {
		mag.Mod(mag, c.base.GetG1Order())
	}
	sc := make([]byte, 32)
	mag.FillBytes(sc)
	return sc, neg
}

func (pt *hipPoint) Mul(k *big.Int) Point {
	sc, neg := pt.c.scalar32(k)
	sign := []byte{byte(neg)}
	out := make([]byte, len(pt.raw))
	if C.bgls_scale_points(pt.c.id, pt.group, p(pt.raw), p(sc), p(sign), 1, p(out)) != 0 {
		return nil
	}
	return &hipPoint{pt.c, pt.group, out}
}
func (pt *hipPoint) ToAffineCoords() []*big.Int {
	n := int(C.bgls_fp_size(pt.c.id))
	r := make([]*big.Int, len(pt.raw)/n)
	for i := range r {
		r[i] = new(big.Int).SetBytes(pt.raw[i*n : (i+1)*n])
	}
	return r
}

// ---- PointT (curves/curve.go:62-70) ------------------------------------------------------
func (t hipPointT) Add(o PointT) (PointT, bool) {
	q, ok := o.(hipPointT)
	if !ok || q.c != t.c {
		return nil, false
	}
	out := make([]byte, len(t.raw))
	if C.bgls_gt_mul(t.c.id, p(t.raw), p(q.raw), p(out)) != 0 {
		return nil, false
	}
	return hipPointT{t.c, out}, true
}
func (t hipPointT) Copy() PointT    { return hipPointT{t.c, append([]byte(nil), t.raw...)} }
func (t hipPointT) Marshal() []byte { return append([]byte(nil), t.raw...) }
func (t hipPointT) Equals(o PointT) bool {
	q, ok := o.(hipPointT)
	return ok && bytes.Equal(q.raw, t.raw)
}
func (t hipPointT) Mul(k *big.Int) PointT {
	sc, neg := t.c.scalar32(k)
	out := make([]byte, len(t.raw))
	if C.bgls_gt_pow(t.c.id, p(t.raw), p(sc), neg, p(out)) != 0 {
		return nil // the interface never panics (curves/curve.go:15-22)
	}
	return hipPointT{t.c, out}
}

// ---- CurveSystem (curves/curve.go:12-49) -------------------------------------------------
func (c *hipCurve) Name() string { return c.name }

func (c *hipCurve) unmarshal(group C.int, data []byte) (Point, bool) {
	if 2*len(data) == c.size(group) { // compressed branch, curves/altbn128.go:296-376, curves/bls12_381.go:242-264
		out := make([]byte, c.size(group))
		ok := []byte{0}
		if C.bgls_decompress_points(c.id, group, p(data), 1, p(out), p(ok)) != 0 || ok[0] != 1 {
			return nil, false
		}
		return &hipPoint{c, group, out}, true
	}
	if len(data) != c.size(group) || C.bgls_point_check(c.id, group, p(data)) != 1 {
		return nil, false
	}
	return &hipPoint{c, group, append([]byte(nil), data...)}, true
}
func (c *hipCurve) UnmarshalG1(d []byte) (Point, bool) { return c.unmarshal(C.BGLS_G1, d) }
func (c *hipCurve) UnmarshalG2(d []byte) (Point, bool) { return c.unmarshal(C.BGLS_G2, d) }
func (c *hipCurve) UnmarshalGT(d []byte) (PointT, bool) {
	if len(d) != int(C.bgls_gt_size(c.id)) {
		return nil, false
	}
	return hipPointT{c, append([]byte(nil), d...)}, true
}
func (c *hipCurve) makePoint(group C.int, coords []*big.Int, check bool) (Point, bool) {
	n := int(C.bgls_fp_size(c.id))
	if len(coords)*n != c.size(group) {
		return nil, false
	}
	raw := make([]byte, len(coords)*n)
	for i, v := range coords {
		if v.Sign() < 0 || v.BitLen() > 8*n {
			return nil, false
		}
		v.FillBytes(raw[i*n : (i+1)*n])
	}
	// bgls_point_check = canonical coordinates, on the curve and (G2) in the order-r subgroup: what upstream's Unmarshal
	// checks.  alt-bn128 validates regardless of `check` (curves/altbn128.go:39-41,157-160), bls12 only with check.
	if !check && c.id != C.BGLS_CURVE_ALTBN128 {
		return &hipPoint{c, group, raw}, true
	}
	return c.unmarshal(group, raw)
}
func (c *hipCurve) MakeG1Point(co []*big.Int, check bool) (Point, bool) { return c.makePoint(C.BGLS_G1, co, check) }
func (c *hipCurve) MakeG2Point(co []*big.Int, check bool) (Point, bool) { return c.makePoint(C.BGLS_G2, co, check) }

func (c *hipCurve) gen(group C.int) Point {
	out := make([]byte, c.size(group))
	C.bgls_generator(c.id, group, p(out))
	return &hipPoint{c, group, out}
}
func (c *hipCurve) GetG1() Point         { return c.gen(C.BGLS_G1) }
func (c *hipCurve) GetG2() Point         { return c.gen(C.BGLS_G2) }
func (c *hipCurve) GetG1Infinity() Point { return &hipPoint{c, C.BGLS_G1, make([]byte, c.size(C.BGLS_G1))} }
func (c *hipCurve) GetG2Infinity() Point { return &hipPoint{c, C.BGLS_G2, make([]byte, c.size(C.BGLS_G2))} }
func (c *hipCurve) GetGT() PointT        { t, _ := c.Pair(c.GetG1(), c.GetG2()); return t }
func (c *hipCurve) GetGTIdentity() PointT {
	out := make([]byte, int(C.bgls_gt_size(c.id)))
	C.bgls_gt_identity(c.id, p(out))
	return hipPointT{c, out}
}

func (c *hipCurve) HashToG1(message []byte) Point {
	off := []C.uint64_t{0, C.uint64_t(len(message))}
	out := make([]byte, c.size(C.BGLS_G1))
	if C.bgls_hash_to_g1(c.id, p(message), &off[0], 1, p(out)) != 0 {
		return nil
	}
	return &hipPoint{c, C.BGLS_G1, out}
}

func (c *hipCurve) Pair(a Point, b Point) (PointT, bool) {
	return c.PairingProduct([]Point{a}, []Point{b})
}

// PairingProduct routes the whole slice through ONE C call instead of concurrentPairingProduct
// (curves/curve.go:125-170).
func (c *hipCurve) PairingProduct(p1 []Point, p2 []Point) (PointT, bool) {
	if len(p1) != len(p2) {
		return nil, false
	}
	g1 := make([]byte, 0, len(p1)*c.size(C.BGLS_G1))
	g2 := make([]byte, 0, len(p2)*c.size(C.BGLS_G2))
	for i := range p1 {
		a, ok1 := p1[i].(*hipPoint)
		b, ok2 := p2[i].(*hipPoint)
		if !ok1 || !ok2 || a.c != c || b.c != c || a.group != C.BGLS_G1 || b.group != C.BGLS_G2 {
			return nil, false
		}
		g1 = append(g1, a.raw...)
		g2 = append(g2, b.raw...)
	}
	out := make([]byte, int(C.bgls_gt_size(c.id)))
	if C.bgls_pairing_product(c.id, p(g1), p(g2), C.size_t(len(p1)), p(out)) != 0 {
		return nil, false
	}
	return hipPointT{c, out}, true
}

func (c *hipCurve) GetG1Q() *big.Int                     { return c.base.GetG1Q() }
func (c *hipCurve) GetG1Order() *big.Int                 { return c.base.GetG1Order() }
func (c *hipCurve) getG1Cofactor() *big.Int              { return c.base.getG1Cofactor() }
func (c *hipCurve) getG1A() *big.Int                     { return c.base.getG1A() }
func (c *hipCurve) getG1B() *big.Int                     { return c.base.getG1B() }
func (c *hipCurve) getFTHashParams() (*big.Int, *big.Int) { return c.base.getFTHashParams() }
func (c *hipCurve) g1XToYSquared(x *big.Int) *big.Int    { return c.base.g1XToYSquared(x) }

// ---- batch fast paths used by package bgls (one cgo call each) ---------------------------

// IsHip reports whether curve is backed by the HIP engine.
func IsHip(curve CurveSystem) bool { _, ok := curve.(*hipCurve); return ok }

// HipVerifyAggregate is what bgls.verifyAggSig (bgls/bgls.go:94-119) calls when curve is a *hipCurve.
func HipVerifyAggregate(curve CurveSystem, aggsig Point, keys []Point, msgs [][]byte, allowDuplicates bool) bool {
	c, ok := curve.(*hipCurve)
	s, ok2 := aggsig.(*hipPoint)
	if !ok || !ok2 || len(keys) != len(msgs) {
		return false
	}
	kb := make([]byte, 0, len(keys)*c.size(C.BGLS_G2))
	for _, k := range keys {
		q, ok := k.(*hipPoint)
		if !ok || q.group != C.BGLS_G2 {
			return false
		}
		kb = append(kb, q.raw...)
	}
	off := make([]C.uint64_t, len(msgs)+1)
	var blob []byte
	for i, m := range msgs {
		off[i] = C.uint64_t(len(blob))
		blob = append(blob, m...)
	}
	off[len(msgs)] = C.uint64_t(len(blob))
	dup := C.int(0)
	if allowDuplicates {
		dup = 1
	}
	return C.bgls_verify_aggregate(c.id, p(s.raw), p(kb), p(blob), &off[0], C.size_t(len(keys)), dup) == 1
}

// HipVerifyMulti is bgls.verifyMultiSignature (bgls/bgls.go:89-92) in one call.
func HipVerifyMulti(curve CurveSystem, aggsig Point, keys []Point, msg []byte) bool {
	c, ok := curve.(*hipCurve)
	s, ok2 := aggsig.(*hipPoint)
	if !ok || !ok2 {
		return false
	}
	kb := make([]byte, 0, len(keys)*c.size(C.BGLS_G2))
	for _, k := range keys {
		q, ok := k.(*hipPoint)
		if !ok || q.group != C.BGLS_G2 {
			return false
		}
		kb = append(kb, q.raw...)
	}
	return C.bgls_verify_multi(c.id, p(s.raw), p(kb), C.size_t(len(keys)), p(msg), C.size_t(len(msg))) == 1
}

// ---- device-resident key sets -------------------------------------------------------------------------------------------
// HipKeySet is a []Point of public keys uploaded once (parsed, validated, resident in HBM, optionally cut over several
// GPUs); the handle is released by a finalizer, as every other Go-owned device resource would be.
type HipKeySet struct {
	c *hipCurve
	h C.bgls_keys_t
	n int
}

// HipUploadKeys uploads keys to `devices` (nil = the default device).  check = true re-runs the construction-time
// validation (subgroup membership included) on the device.
func HipUploadKeys(curve CurveSystem, keys []Point, devices []int, check bool) *HipKeySet {
	c, ok := curve.(*hipCurve)
	if !ok {
		return nil
	}
	kb, ok2 := hipKeyBytes(c, keys)
	if !ok2 {
		return nil
	}
	devs := make([]C.int, 0, len(devices))
	for _, d := range devices {
		devs = append(devs, C.int(d))
	}
	var dp *C.int
	nd := C.int(1)
	if len(devs) > 0 {
		dp, nd = &devs[0], C.int(len(devs))
	}
	flags := C.uint(0)
	if check {
		flags = C.BGLS_KEYS_CHECK
	}
	var h C.bgls_keys_t
	if C.bgls_keys_upload(c.id, p(kb), C.size_t(len(keys)), dp, nd, flags, &h) != 0 {
		return nil
	}
	ks := &HipKeySet{c, h, len(keys)}
	runtime.SetFinalizer(ks, func(k *HipKeySet) { C.bgls_keys_free(k.h) })
	return ks
}

// VerifyAggregate is bgls.verifyAggSig (bgls/bgls.go:94-119) against the resident keys, on every GPU of the set.
func (ks *HipKeySet) VerifyAggregate(aggsig Point, msgs [][]byte, allowDuplicates bool) bool {
	s, ok := aggsig.(*hipPoint)
	if !ok || s.c != ks.c || s.group != C.BGLS_G1 || len(msgs) != ks.n {
		return false
	}
	off := make([]C.uint64_t, len(msgs)+1)
	var blob []byte
	for i, m := range msgs {
		off[i] = C.uint64_t(len(blob))
		blob = append(blob, m...)
	}
	off[len(msgs)] = C.uint64_t(len(blob))
	dup := C.int(0)
	if allowDuplicates {
		dup = 1
	}
	ok = C.bgls_verify_aggregate_h(ks.h, p(s.raw), p(blob), &off[0], C.size_t(ks.n), dup) == 1
	runtime.KeepAlive(ks)
	return ok
}

// VerifyMulti is bgls.verifyMultiSignature (bgls/bgls.go:89-92) against the resident keys.
func (ks *HipKeySet) VerifyMulti(aggsig Point, msg []byte) bool {
	s, ok := aggsig.(*hipPoint)
	if !ok || s.c != ks.c || s.group != C.BGLS_G1 {
		return false
	}
	ok = C.bgls_verify_multi_h(ks.h, p(s.raw), p(msg), C.size_t(len(msg))) == 1
	runtime.KeepAlive(ks)
	return ok
}

func hipKeyBytes(c *hipCurve, keys []Point) ([]byte, bool) {
	kb := make([]byte, 0, len(keys)*c.size(C.BGLS_G2))
	for _, k := range keys {
		q, ok := k.(*hipPoint)
		if !ok || q.group != C.BGLS_G2 {
			return nil, false
		}
		kb = append(kb, q.raw...)
	}
	return kb, true
}

// HipHashPubKeysToExponents is bgls.hashPubKeysToExponents (bgls/blsHAE.go:80-93): BLAKE2Xb over the keys'
// uncompressed bytes, n 16-byte big-endian exponents.
func HipHashPubKeysToExponents(curve CurveSystem, pubkeys []Point) []*big.Int {
	c, ok := curve.(*hipCurve)
	kb, ok2 := hipKeyBytes(c, pubkeys)
	if !ok || !ok2 || len(pubkeys) == 0 {
		return nil
	}
	out := make([]byte, 16*len(pubkeys))
	if C.bgls_hae_exponents(c.id, p(kb), C.size_t(len(pubkeys)), p(out)) != 0 {
		return nil
	}
	t := make([]*big.Int, len(pubkeys))
	for i := range t {
		t[i] = new(big.Int).SetBytes(out[16*i : 16*i+16])
	}
	return t
}

// HipVerifyMultiHAE is bgls.VerifyMultiSignatureWithHAE (bgls/blsHAE.go:56-58) in one call.
func HipVerifyMultiHAE(curve CurveSystem, aggsig Point, pubkeys []Point, msg []byte) bool {
	c, ok := curve.(*hipCurve)
	s, ok2 := aggsig.(*hipPoint)
	if !ok || !ok2 {
		return false
	}
	kb, ok3 := hipKeyBytes(c, pubkeys)
	if !ok3 {
		return false
	}
	return C.bgls_verify_multi_hae(c.id, p(s.raw), p(kb), C.size_t(len(pubkeys)), p(msg), C.size_t(len(msg))) == 1
}

// HipVerifyAggregateHAE is bgls.VerifyAggregateSignatureWithHAE (bgls/blsHAE.go:49-53) in one call.
func HipVerifyAggregateHAE(curve CurveSystem, aggsig Point, pubkeys []Point, msgs [][]byte) bool {
	c, ok := curve.(*hipCurve)
	s, ok2 := aggsig.(*hipPoint)
	if !ok || !ok2 || len(pubkeys) != len(msgs) {
		return false
	}
	kb, ok3 := hipKeyBytes(c, pubkeys)
	if !ok3 {
		return false
	}
	off := make([]C.uint64_t, len(msgs)+1)
	var blob []byte
	for i, m := range msgs {
		off[i] = C.uint64_t(len(blob))
		blob = append(blob, m...)
	}
	off[len(msgs)] = C.uint64_t(len(blob))
	return C.bgls_verify_aggregate_hae(c.id, p(s.raw), p(kb), p(blob), &off[0], C.size_t(len(pubkeys))) == 1
}

// HipAggregateSignaturesHAE is bgls.AggregateSignaturesWithHAE (bgls/blsHAE.go:39-46) in one call.
func HipAggregateSignaturesHAE(curve CurveSystem, sigs []Point, pubkeys []Point) Point {
	c, ok := curve.(*hipCurve)
	if !ok || len(sigs) != len(pubkeys) || len(sigs) == 0 {
		return nil
	}
	kb, ok2 := hipKeyBytes(c, pubkeys)
	if !ok2 {
		return nil
	}
	sb := make([]byte, 0, len(sigs)*c.size(C.BGLS_G1))
	for _, x := range sigs {
		q, ok := x.(*hipPoint)
		if !ok || q.group != C.BGLS_G1 {
			return nil
		}
		sb = append(sb, q.raw...)
	}
	out := make([]byte, c.size(C.BGLS_G1))
	if C.bgls_aggregate_signatures_hae(c.id, p(sb), p(kb), C.size_t(len(sigs)), p(out)) != 0 {
		return nil
	}
	return &hipPoint{c, C.BGLS_G1, out}
}

// HipVerifyMultiWithMultiplicity is the body of bgls.KoskVerifyMultiSignatureWithMultiplicity (bgls/blsKosk.go:137-150);
// msg must already carry the Kosk 0x01 prefix.
func HipVerifyMultiWithMultiplicity(curve CurveSystem, aggsig Point, keys []Point, multiplicity []int64, msg []byte) bool {
	c, ok := curve.(*hipCurve)
	s, ok2 := aggsig.(*hipPoint)
	if !ok || !ok2 || len(keys) != len(multiplicity) || len(keys) == 0 {
		return false
	}
	kb, ok3 := hipKeyBytes(c, keys)
	if !ok3 {
		return false
	}
	return C.bgls_verify_multi_multiplicity(c.id, p(s.raw), p(kb), (*C.int64_t)(&multiplicity[0]), C.size_t(len(keys)), p(msg), C.size_t(len(msg))) == 1
}

// HipVerifyBatchMulti is the body of bgls.KoskVerifyBatchMultiSignature (bgls/blsKosk.go:126-133) in one call: the key sums of
// all sets in one launch, then one aggregate verification over len(msgs) pairs.  msgs must already carry the Kosk 0x01 prefix.
func HipVerifyBatchMulti(curve CurveSystem, aggsigs []Point, pubkeys [][]Point, msgs [][]byte) bool {
	c, ok := curve.(*hipCurve)
	if !ok || len(aggsigs) != len(pubkeys) || len(pubkeys) != len(msgs) || len(msgs) == 0 {
		return false
	}
	sb := make([]byte, 0, len(aggsigs)*int(c.size(C.BGLS_G1)))
	var kb, blob []byte
	koff := make([]C.uint64_t, len(msgs)+1)
	moff := make([]C.uint64_t, len(msgs)+1)
	for i := range msgs {
		s, isHip := aggsigs[i].(*hipPoint)
		if !isHip || s.group != C.BGLS_G1 {
			return false
		}
		sb = append(sb, s.raw...)
		one, ok3 := hipKeyBytes(c, pubkeys[i])
		if !ok3 {
			return false
		}
		kb = append(kb, one...)
		koff[i+1] = koff[i] + C.uint64_t(len(pubkeys[i]))
		blob = append(blob, msgs[i]...)
		moff[i+1] = C.uint64_t(len(blob))
	}
	return C.bgls_verify_multi_batch(c.id, p(sb), p(kb), &koff[0], C.size_t(len(msgs)), p(blob), &moff[0], 1) == 1
}

// HipUnmarshalG2Batch decodes n compressed or uncompressed alt-bn128 keys in one call (the batch form of UnmarshalG2 for
// key sets arriving over the wire); ok[i] mirrors the per-point (Point, bool).
func HipUnmarshalG2Batch(curve CurveSystem, data []byte, n int) ([]Point, []bool) {
	c, isHip := curve.(*hipCurve)
	if !isHip || n == 0 || len(data)%n != 0 {
		return nil, nil
	}
	sz := c.size(C.BGLS_G2)
	raw := data
	oks := make([]byte, n)
	if len(data)/n*2 == sz {
		raw = make([]byte, n*sz)
		if C.bgls_decompress_points(c.id, C.BGLS_G2, p(data), C.size_t(n), p(raw), p(oks)) != 0 {
			return nil, nil
		}
	} else {
		if C.bgls_check_points(c.id, C.BGLS_G2, p(raw), C.size_t(n), p(oks)) != 0 {
			return nil, nil
		}
	}
	pts := make([]Point, n)
	good := make([]bool, n)
	for i := range pts {
		if oks[i] == 1 {
			pts[i] = &hipPoint{c, C.BGLS_G2, append([]byte(nil), raw[i*sz:(i+1)*sz]...)}
			good[i] = true
		}
	}
	return pts, good
}
