#!/usr/bin/env python3
"""Generate bgls_amd/csrc/constants_gen.hpp: per-curve constants as little-endian u32 limbs.

Stand-alone on purpose (imports nothing from oracle/ -- the product build must not depend on the
checker).  Everything is derived from the curve-family polynomials and the literal constants the
reference lists in curves/altbn128.go:458-480 and curves/bls12_381.go:328-346.
Field elements are emitted in Montgomery form (R = 2^(32 L)); exponents/scalars as plain integers.
"""
import os, sys

OUT = os.path.join(os.path.dirname(__file__), "..", "bgls_amd", "csrc", "constants_gen.hpp")


def limbs(x, n):
    assert 0 <= x < (1 << (32 * n)), (x, n)
    return [(x >> (32 * i)) & 0xFFFFFFFF for i in range(n)]


def arr(name, vals, ctype="uint32_t"):
    return "  static constexpr %s %s[%d] = {%s};\n" % (ctype, name, len(vals), ", ".join("0x%08xu" % v for v in vals))


def naf(k):
    out = []
    while k:
        d = (2 - (k % 4)) if (k & 1) else 0
        k -= d
        out.append(d)
        k >>= 1
    return out[::-1]


class F2:
    def __init__(s, p): s.p = p
    def mul(s, a, b): p = s.p; return ((a[0]*b[0]-a[1]*b[1]) % p, (a[0]*b[1]+a[1]*b[0]) % p)
    def inv(s, a):
        p = s.p; n = pow(a[0]*a[0]+a[1]*a[1], p-2, p); return (a[0]*n % p, -a[1]*n % p)
    def pow(s, a, e):
        r = (1, 0)
        while e:
            if e & 1: r = s.mul(r, a)
            a = s.mul(a, a); e >>= 1
        return r


def emit(cname, cid, L, p, r, b, xi, twist, loop, extra):
    R = 1 << (32 * L)
    M = lambda x: (x % p) * R % p
    f2 = F2(p)
    o = "struct %s {\n" % cname
    o += "  static constexpr int L = %d;\n  static constexpr int CURVE_ID = %d;\n" % (L, cid)
    o += "  static constexpr bool TWIST_D = %s;\n" % ("true" if twist == "D" else "false")
    o += "  static constexpr int FP_BYTES = %d;\n" % (4 * L)
    o += "  static constexpr uint32_t N0INV = 0x%08xu;\n" % ((-pow(p, -1, 1 << 32)) % (1 << 32))
    o += "  static constexpr int XI_RE = %d;\n" % xi[0]
    o += arr("P", limbs(p, L))
    o += arr("P2W", limbs(p * p, 2 * L))          # p^2, for lazy-reduction offsets
    o += arr("P2W3", limbs(3 * p * p, 2 * L))
    o += arr("P2W6", limbs(6 * p * p, 2 * L))
    o += arr("ONE", limbs(M(1), L))
    o += arr("R2", limbs(R * R % p, L))
    o += arr("HALF", limbs(M(pow(2, p - 2, p)), L))
    o += arr("B", limbs(M(b), L))
    b2 = f2.mul((b, 0), f2.inv(xi)) if twist == "D" else f2.mul((b, 0), xi)
    o += arr("B2_RE", limbs(M(b2[0]), L)) + arr("B2_IM", limbs(M(b2[1]), L))
    b23 = (3 * b2[0] % p, 3 * b2[1] % p)
    o += arr("B2X3_RE", limbs(M(b23[0]), L)) + arr("B2X3_IM", limbs(M(b23[1]), L))
    # Frobenius constants gamma_j[k] = xi^(k (p^j - 1)/6), flattened [j-1][k][re,im][L]
    g = []
    for j in (1, 2, 3):
        g1 = f2.pow(xi, (p ** j - 1) // 6)
        cur = (1, 0)
        for k in range(6):
            g += limbs(M(cur[0]), L) + limbs(M(cur[1]), L)
            cur = f2.mul(cur, g1)
    o += arr("GAMMA", g)
    # psi = twist o Frobenius o untwist on E'(Fp2): psi(x, y) = (conj(x) PSI_X, conj(y) PSI_Y) with
    # PSI_X = xi^((p-1)/3), PSI_Y = xi^((p-1)/2) on a D-type twist and their inverses on an M-type twist (subgroup test)
    px, py = f2.pow(xi, (p - 1) // 3), f2.pow(xi, (p - 1) // 2)
    if twist != "D":
        px, py = f2.inv(px), f2.inv(py)
    o += arr("PSI_X", limbs(M(px[0]), L) + limbs(M(px[1]), L)) + arr("PSI_Y", limbs(M(py[0]), L) + limbs(M(py[1]), L))
    o += arr("EXP_SQRT", limbs((p + 1) // 4, L))   # calcQuadRes exponent, hash.go:178-190
    o += arr("EXP_INV", limbs(p - 2, L))
    o += arr("ORDER", limbs(r, 8))
    digs = naf(loop)
    o += "  static constexpr int LOOP_LEN = %d;\n" % len(digs)
    o += "  static constexpr int8_t LOOP_NAF[%d] = {%s};\n" % (len(digs), ", ".join(str(d) for d in digs))
    o += extra(M, limbs, L)
    o += "};\n\n"
    return o


def main():
    u = 4965661367192848881
    p_bn = 36*u**4 + 36*u**3 + 24*u**2 + 6*u + 1
    r_bn = 36*u**4 + 36*u**3 + 18*u**2 + 6*u + 1
    assert p_bn == 21888242871839275222246405745257275088696311157297823662689037894645226208583
    x = -0xd201000000010000
    p_bls = (x - 1)**2 * (x**4 - x**2 + 1) // 3 + x
    r_bls = x**4 - x**2 + 1
    assert p_bls == 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab

    def r28_consts(p, L):
        """Constants of the 28-bit-limb consumer representation (r28.hpp): R' = 2^280, 10 limbs."""
        W, N = 28, 10
        Mk = (1 << W) - 1
        lim = lambda x: [(x >> (W * i)) & Mk for i in range(N)]
        R = 1 << (32 * L)
        Rp = 1 << (W * N)
        o = ""
        o += arr("R28_P", lim(p))
        o += "  static constexpr uint32_t R28_NP = 0x%xu;\n" % ((-pow(p, -1, 1 << W)) % (1 << W))
        # "fat" multiple of p for limb-wise negation of a tight operand: limbs 0..8 in [2^28, 2^29), top limb = what is left
        k = 64
        base = sum((1 << 28) << (W * i) for i in range(N - 1))
        d = k * p - base
        assert d >= 0
        fat = [(1 << 28) + ((d >> (W * i)) & Mk) for i in range(N - 1)] + [d >> (W * (N - 1))]
        assert sum(v << (W * i) for i, v in enumerate(fat)) == k * p and all(v < (1 << 29) for v in fat[:-1])
        o += arr("R28_FAT", fat)
        o += arr("R28_ONE", lim(Rp % p))                              # 1 in R' form
        o += "  static constexpr uint32_t R28_MU = 0x%xu;\n" % ((1 << 278) // p)   # Barrett: q = (top32(y) * MU) >> 32 ~ y 2^24 / p
        # Column biases for the Karatsuba form of a three-term Fp2 dot product (coop_r28.hpp): column k of
        # sum (a0+a1)(b0+b1) - sum a0 b0 - sum a1 b1 and of sum a0 b0 - sum a1 b1 can be negative even though the totals are
        # not; BIAS3[k] >= the largest possible sum (a0 b0 + a1 b1)[k] over three terms of tight operands (limbs < 2^28, top
        # limb < 2^12), and the whole array is a multiple of p, so adding it changes nothing mod p.
        la = [1 << 28] * 9 + [1 << 12]
        col = [0] * 20
        for i in range(10):
            for j in range(10):
                col[i + j] += (la[i] - 1) * (la[j] - 1)
        bias = []
        for k in range(20):
            need = 2 * 3 * col[k]
            b = 1
            while b < need + 1:
                b <<= 1
            bias.append(b if need else 0)
        Vb = sum(b << (W * k) for k, b in enumerate(bias))
        fix = (-Vb) % p
        for k in range(10):
            bias[k] += (fix >> (W * k)) & Mk
        assert fix >> (W * 10) == 0 and sum(b << (W * k) for k, b in enumerate(bias)) % p == 0
        scol = [0] * 20
        for i in range(10):
            for j in range(10):
                scol[i + j] += (2 * la[i] - 1) * (2 * la[j] - 1)
        assert all(3 * scol[k] + bias[k] < (1 << 64) for k in range(20))
        o += "  static constexpr uint64_t R28_BIAS3[20] = {%s};\n" % ", ".join("0x%xull" % b for b in bias)
        o += arr("R28_BACK", limbs((R * R // Rp) % p * 1 % p if (R * R) % Rp == 0 else (R * R * pow(Rp, -1, p)) % p, L))   # 32-bit Montgomery multiplier: x R' -> x R
        return o

    def rx_consts(p, L, N, xi, b2x3=None, W=28, VB=4, VBND=32):
        """Constants of the generic carry-free W-bit-limb representation (rx.hpp): N limbs, Montgomery radix R' = 2^(W N).
        alt-bn128: N = 10, W = 28 (26 spare bits), BLS12-381: N = 14, W = 28 (11 spare bits); round 5: alt-bn128's Miller kernel
        on N = 9, W = 29 (7.4 spare bits: struct BN254W) -- 81 instead of 100 multiplier instructions per limb product.
        VB: a signed value handed to sx_to_ux must lie above -K VB p; VBND: value bound (multiples of p) of the consumer's operands."""
        Mk = (1 << W) - 1
        lim = lambda x: [(x >> (W * i)) & Mk for i in range(N)]
        R = 1 << (32 * L)
        Rp = 1 << (W * N)
        assert Rp > 4 * p
        o = "  static constexpr int RX_NL = %d;\n  static constexpr int RX_W = %d;\n  static constexpr uint32_t RX_MASK = 0x%xu;\n" % (N, W, Mk)
        o += "  static constexpr int RX_VBND = %d;\n" % VBND
        o += arr("RX_P", lim(p))
        o += "  static constexpr uint32_t RX_NP = 0x%xu;\n" % ((-pow(p, -1, 1 << W)) % (1 << W))
        o += arr("RX_ONE", lim(Rp % p))
        o += arr("RX_R2", lim(Rp * Rp % p))
        # (x R mod p, as an integer in 28-bit limbs) * RX_TO / R' = x R'
        o += arr("RX_TO", lim(Rp * Rp * pow(R, -1, p) % p))
        # 32-bit Montgomery multiplier: (x R' mod p as an integer) * BACK / R = x R
        o += arr("RX_BACK", limbs(R * R * pow(Rp, -1, p) % p, L))
        # round 6: (x R' in the carry-free form) * RX_TOM / R' = x R as a PLAIN integer in W-bit limbs: the library's 32-bit Montgomery residue after
        # one carry-free product and a repacking (from_ux_inl pays a 32-bit Montgomery product for the same conversion)
        o += arr("RX_TOM", lim(R % p))
        # Fat multiples of p for limb-wise negation: FAT[k-1] has limbs 0..N-2 in [k 2^28, (k+1) 2^28) and a top limb >= TOPK * k,
        # so FAT_k - b has non-negative limbs for every b with limbs below k 2^28 and value below k * RX_FAT_VB * p
        top_p = p >> (W * (N - 1))
        fats = []
        KMAX = 8 if W <= 28 else 6                 # (k + 1) 2^W must stay below 2^32
        for k in range(1, KMAX + 1):
            base = sum((k << W) << (W * i) for i in range(N - 1)) + ((k * VB * (top_p + 1) + 1) << (W * (N - 1)))
            mult = -(-base // p)
            d = mult * p - base
            assert 0 <= d < p
            assert mult <= 2 * k * VB + 2
            f = [(k << W) + ((d >> (W * i)) & Mk) for i in range(N - 1)] + [(k * VB * (top_p + 1) + 1) + (d >> (W * (N - 1)))]
            assert sum(v << (W * i) for i, v in enumerate(f)) == mult * p and all((k << W) <= v < ((k + 1) << W) for v in f[:-1])
            assert f[-1] < (1 << 31)
            fats += f
        o += "  static constexpr int RX_FAT_VB = %d;\n  static constexpr int RX_FAT_KMAX = %d;\n" % (VB, KMAX)
        o += arr("RX_FAT", fats)                   # [8][N]
        # Column biases (multiples of p) for the consumer's dot products of TIGHT non-negative operands (limbs < 2^28, value < 32 p):
        #   RX_BIAS_D3: >= column k of sum_{t<3} a1 b1           (re = D + BIAS - E of the three-term Karatsuba fold)
        #   RX_BIAS_S6: >= column k of six term-equivalents      (symmetric squaring: up to 3 doubled + 0 plain or 2 doubled + 2 plain)
        la = [1 << W] * (N - 1) + [VBND * (top_p + 1)]
        assert la[-1] <= (1 << W)
        col = [0] * (2 * N)
        for i in range(N):
            for j in range(N):
                col[i + j] += (la[i] - 1) * (la[j] - 1)
        def bias_for(mult):
            bias = []
            for k in range(2 * N):
                need = mult * col[k]
                bias.append(((need >> W) + 1) << W if need else 0)      # multiple of 2^28 above the need: leaves the low digit free for the fix-up
            Vb = sum(b << (W * k) for k, b in enumerate(bias))
            fix = (-Vb) % p
            for k in range(N):
                bias[k] += (fix >> (W * k)) & Mk
            assert fix >> (W * N) == 0 and sum(b << (W * k) for k, b in enumerate(bias)) % p == 0
            assert all(bias[k] >= mult * col[k] for k in range(2 * N))
            return bias
        b3 = bias_for(3)
        lazy = 2 * W + 8 <= 64                     # 2^8 of head-room per column (W = 28); W = 29 has 2^6: three-term piles only
        b6 = bias_for(6) if lazy else [0] * (2 * N)
        # 64-bit column budget: worst case of every pile the consumer forms, plus the reduction's own products and carries
        red = [0] * (2 * N)
        for i in range(N):
            for j in range(N):
                red[i + j] += Mk * Mk
        scol = [0] * (2 * N)
        for i in range(N):
            for j in range(N):
                scol[i + j] += (2 * la[i] - 2) * (2 * la[j] - 2)
        for k in range(2 * N):
            carry = 1 << 37
            assert 3 * col[k] + b3[k] + red[k] + carry < (1 << 64), ("D + BIAS", k)
            # the cross pile is formed as -(D + E) + sum (a0 + a1)(b0 + b1) mod 2^64: its FINAL totals sum (a0 b1 + a1 b0) are what must fit
            assert 6 * col[k] + red[k] + carry < (1 << 64), ("cross pile", k)
            if lazy:
                assert 3 * scol[k] + red[k] + carry < (1 << 64), ("S pile", k)             # sum (a0+a1)(b0+b1), three terms, formed alone by r28-style callers
                assert 6 * col[k] + b6[k] + red[k] + carry < (1 << 64), ("sqr D + BIAS", k)
                # six term-equivalents x (a0 b1 + a1 b0): the FINAL totals of the squaring's cross pile, which rx.hpp's ux_sqr_dot forms as
                # -(D + E) + sum (a0 + a1)(b0 + b1) mod 2^64 (Karatsuba; the sum alone would not fit on 14 limbs, it is never formed alone)
                assert 12 * col[k] + red[k] + carry < (1 << 64), ("sqr cross pile", k)
        o += "  static constexpr uint64_t RX_BIAS_D3[%d] = {%s};\n" % (2 * N, ", ".join("0x%xull" % b for b in b3))
        o += "  static constexpr uint64_t RX_BIAS_S6[%d] = {%s};\n" % (2 * N, ", ".join("0x%xull" % b for b in b6))
        if b2x3 is not None:
            o += arr("RX_B2X3_RE", lim(b2x3[0] * Rp % p)) + arr("RX_B2X3_IM", lim(b2x3[1] * Rp % p))
            inv3 = pow(3, -1, p)
            o += arr("RX_B2_RE", lim(b2x3[0] * inv3 % p * Rp % p)) + arr("RX_B2_IM", lim(b2x3[1] * inv3 % p * Rp % p))   # b' of the twist
        f2x = F2(p)
        gam = []
        for (j, k) in ((1, 2), (1, 3), (2, 2), (2, 3)):          # the Frobenius constants of the two extra alt-bn128 line steps (pairing.hpp)
            gv = f2x.pow(xi, k * (p ** j - 1) // 6)
            gam += lim(gv[0] * Rp % p) + lim(gv[1] * Rp % p)
        o += arr("RX_GAMMA", gam)                                 # [(1,2), (1,3), (2,2), (2,3)][re, im][N]
        o += arr("RX_PK", sum((lim(k * p) for k in range(9)), []))    # tight limbs of 0, p, 2p .. 8p (exact zero test of a lazy value)
        o += arr("RX_LAD", sum((lim((1 << k) * p) for k in range(1, 7)), []))   # tight limbs of 2p, 4p .. 64p (ux_quasi: conditional subtractions)
        if not lazy:
            # ux_mulxi on the narrow form subtracts an estimated quotient q < 32 in the same pass: xi a - q p = xi a + (32 - q) p + G with
            # G0 = (fat 4p) - 32 p for the real part (low limbs in [2^W, 2^(W+1)): they dominate the subtracted a1; signed top limb) and G1 = -32 p
            def signed_rep(v, lowbase):
                base = sum(lowbase << (W * i) for i in range(N - 1))
                d = (v - base) % (1 << (W * (N - 1)))
                low = [lowbase + ((d >> (W * i)) & Mk) for i in range(N - 1)]
                rest = v - sum(x << (W * i) for i, x in enumerate(low))
                assert rest % (1 << (W * (N - 1))) == 0
                top = rest >> (W * (N - 1))
                assert -(1 << 31) < top < (1 << 31) and sum(x << (W * i) for i, x in enumerate(low)) + (top << (W * (N - 1))) == v
                return low + [top]
            g0 = signed_rep(4 * p - 32 * p, 1 << W)          # 4 p: the subtracted a1 may be anything below 4 p
            g1 = signed_rep(-32 * p, 0)
            for nm, g in (("RX_XIG0", g0), ("RX_XIG1", g1)):
                o += "  static constexpr int64_t %s[%d] = {%s};\n" % (nm, N, ", ".join("%dll" % v for v in g))
        return o

    def bn_extra(M, limbs, L):
        g2 = [10857046999023057135944570762232829481370756359578518086990519993285655852781,
              11559732032986387107991004021392285783925812861821192530917403151452391805634,
              8495653923123431417604973247489272438418190587263600148770280649306958101930,
              4082367875863433681332203403145435568316851327593401208105741076214120093531]
        o = arr("G1X", limbs(M(1), L)) + arr("G1Y", limbs(M(2), L))
        o += arr("G2", sum((limbs(M(v), L) for v in g2), []))      # x_re, x_im, y_re, y_im
        o += arr("U_ABS", limbs(u, 2))
        o += "  static constexpr int U_BITS = %d;\n" % u.bit_length()
        return o

    def bls_extra(M, limbs, L):
        g1x = 0x17f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb
        g1y = 0x08b3f481e3aaa0f1a09e30ed741d8ae4fcf5e095d5d00af600db18cb2c04b3edd03cc744a2888ae40caa232946c5e7e1
        g2 = [0x024aa2b2f08f0a91260805272dc51051c6e47ad4fa403b02b4510b647ae3d1770bac0326a805bbefd48056c8c121bdb8,
              0x13e02b6052719f607dacd3a088274f65596bd0d09920b61ab5da61bbdc7f5049334cf11213945d57e5ac7d055d042b7e,
              0x0ce5d527727d6e118cc9cdc6da2e351aadfd9baa8cbdd3a76d429a695160d12c923ac9cc3baca289e193548608b82801,
              0x0606c4a02ea734cc32acd2b02bc28b99cb3e287e85a763af267492ab572e99ab3f370d275cec1da1aaa9075ff05f79be]
        sqrt_m3 = 1586958781458431025242759403266842894121773480562120986020912974854563298150952611241517463240701
        z_sw = 793479390729215512621379701633421447060886740281060493010456487427281649075476305620758731620350
        root1 = 248294325734266649657405162895821171812231848760181225578082735178502750823719347628762635478508544819911854747095
        cof = (x - 1)**2 // 3
        assert cof == 76329603384216526031706109802092473003
        o = arr("G1X", limbs(M(g1x), L)) + arr("G1Y", limbs(M(g1y), L))
        o += arr("G2", sum((limbs(M(v), L) for v in g2), []))
        o += arr("U_ABS", limbs(-x, 2))
        o += "  static constexpr int U_BITS = %d;\n" % (-x).bit_length()
        o += arr("SQRT_M3", limbs(M(sqrt_m3), L)) + arr("Z_SW", limbs(M(z_sw), L))
        o += arr("FT_ROOT1", limbs(root1, L)) + arr("FT_ROOT2", limbs(p_bls - root1, L))   # plain (compared before Montgomery conversion)
        o += arr("COFACTOR", limbs(cof, 4))
        o += "  static constexpr int COFACTOR_BITS = %d;\n" % cof.bit_length()
        cn = naf(cof)
        o += "  static constexpr int COFACTOR_NAF_LEN = %d;\n" % len(cn)
        o += "  static constexpr int8_t COFACTOR_NAF[%d] = {%s};\n" % (len(cn), ", ".join(str(d) for d in cn))
        # G1K = (cof^-1 mod r) * g1: the verification path pairs UNCLEARED hash points and raises the product to
        # the cofactor in GT instead (e(h S, Q) = e(S, Q)^h on all of E(Fp)), so the rare "+-generator" outcomes
        # of the hash (curves/bls12_381.go:197-216) enter as +-G1K.
        def ec_add(P, Q):
            if P is None: return Q
            if Q is None: return P
            if P[0] == Q[0]:
                if (P[1] + Q[1]) % p_bls == 0: return None
                lam = 3 * P[0] * P[0] * pow(2 * P[1], -1, p_bls) % p_bls
            else:
                lam = (Q[1] - P[1]) * pow(Q[0] - P[0], -1, p_bls) % p_bls
            x3 = (lam * lam - P[0] - Q[0]) % p_bls
            return (x3, (lam * (P[0] - x3) - P[1]) % p_bls)
        def ec_mul(P, k):
            R = None
            for bit in bin(k)[2:]:
                R = ec_add(R, R)
                if bit == "1": R = ec_add(R, P)
            return R
        assert ec_mul((g1x, g1y), r_bls) is None
        kinv = pow(cof, -1, r_bls)
        g1k = ec_mul((g1x, g1y), kinv)
        assert ec_mul(g1k, cof) == (g1x, g1y)
        o += arr("G1KX", limbs(M(g1k[0]), L)) + arr("G1KY", limbs(M(g1k[1]), L))
        o += arr("R3", limbs((1 << (32 * L)) ** 3 % p_bls, L))    # to Montgomery-convert a 2L-limb value: redc(wide) * R3
        # the Shallue-van de Woestijne constants in the carry-free form (k_bls_sw_jacobi's fractions, round 6): 1 + b, Z = (-1 + sqrt(-3)) / 2, sqrt(-3), times R' = 2^392
        Rp = 1 << (28 * 14)
        l28 = lambda v: [(v >> (28 * i)) & ((1 << 28) - 1) for i in range(14)]
        o += arr("RX_SW_U0", l28(5 * Rp % p_bls)) + arr("RX_SW_Z", l28(z_sw * Rp % p_bls)) + arr("RX_SW_S3", l28(sqrt_m3 * Rp % p_bls))
        # t = (lo + hi 2^384) mod p from the 512-bit digest in one two-product reduction: lo R'^2 + hi (R'^2 2^384), over R'
        o += arr("RX_SW_H384", l28(Rp * Rp * (1 << 384) % p_bls))
        return o

    f2bn = F2(p_bn)
    b2bn = f2bn.mul((3, 0), f2bn.inv((9, 1)))
    bn_b2x3 = (3 * b2bn[0] % p_bn, 3 * b2bn[1] % p_bn)
    txt = "// GENERATED by tools/gen_constants.py -- do not edit.\n#pragma once\n#include <stdint.h>\n\nnamespace bgls {\n\n"
    txt += emit("BN254", 0, 8, p_bn, r_bn, 3, (9, 1), "D", 6 * u + 2, lambda M, limbs, L: bn_extra(M, limbs, L) + r28_consts(p_bn, L) + rx_consts(p_bn, L, 10, (9, 1), bn_b2x3))
    txt += emit("BLS381", 1, 12, p_bls, r_bls, 4, (1, 1), "M", -x, lambda M, limbs, L: bls_extra(M, limbs, L) + rx_consts(p_bls, L, 14, (1, 1), (12, 12)))
    # alt-bn128 on nine 29-bit limbs: the form of the Miller kernel k_miller_x60 alone (round 5).  Everything else of the curve is inherited.
    txt += "struct BN254W : BN254 {\n" + rx_consts(p_bn, 8, 9, (9, 1), bn_b2x3, W=29, VB=1, VBND=4) + "};\n\n"
    txt += "}  // namespace bgls\n"
    with open(OUT, "w") as f:
        f.write(txt)
    print("wrote", os.path.normpath(OUT), len(txt), "bytes")

    # ---- constants of the latency Miller kernel alone (k_millerlatx.hip), in a header of their own so that a change here does
    # not rebuild every unit: beta with 3 b' = xi beta^2 (b' = the twist's constant) and xi beta, R' form of rx.hpp.  The kernel
    # carries Zt = beta Z and Zu = xi beta Z beside Z, so that the doubling step's E = 3 b' Z^2 = Zu Zt is one of the products of
    # the FIRST round instead of a product by a constant in a round of its own.
    def sqrt_fp(a, p):
        assert p % 4 == 3
        s = pow(a, (p + 1) // 4, p)
        return s if s * s % p == a % p else None
    def lat(cname, p, N, xi, b2x3, beta):
        f2 = F2(p)
        assert f2.mul(xi, f2.mul(beta, beta)) == (b2x3[0] % p, b2x3[1] % p), cname
        Rp = 1 << (28 * N)
        lim = lambda x: [(x >> (28 * i)) & ((1 << 28) - 1) for i in range(N)]
        o = "template <>\nstruct LatxK<%s> {\n" % cname
        o += arr("BETA_RE", lim(beta[0] * Rp % p)) + arr("BETA_IM", lim(beta[1] * Rp % p))
        bx = f2.mul(xi, beta)
        o += arr("XIBETA_RE", lim(bx[0] * Rp % p)) + arr("XIBETA_IM", lim(bx[1] * Rp % p))
        # the Frobenius constants of finalx.hpp (fx_frob) in this form: gamma_j[k] = xi^(k (p^j - 1)/6), [j-1][k][re, im][N] -- the
        # same values as GAMMA of constants_gen.hpp, whose 32-bit Montgomery form cost two conversions per coefficient and map
        g = []
        for j in (1, 2, 3):
            g1 = f2.pow(xi, (p ** j - 1) // 6)
            cur = (1, 0)
            for k in range(6):
                g += lim(cur[0] * Rp % p) + lim(cur[1] * Rp % p)
                cur = f2.mul(cur, g1)
        o += arr("FROB_GAMMA", g)
        return o + "};\n"
    xinv = f2bn.inv((9, 1))
    beta_bn = (3 * xinv[0] % p_bn, 3 * xinv[1] % p_bn)                       # 9 / xi = xi (3 / xi)^2
    s12 = sqrt_fp(12, p_bls)
    beta_bls = (s12, 0) if s12 is not None else (0, sqrt_fp(-12 % p_bls, p_bls))      # 12 xi = xi sqrt(12)^2
    t2 = "// GENERATED by tools/gen_constants.py -- do not edit.\n#pragma once\n#include \"constants_gen.hpp\"\n\nnamespace bgls {\n\n"
    t2 += "template <class C>\nstruct LatxK;\n"
    t2 += lat("BN254", p_bn, 10, (9, 1), bn_b2x3, beta_bn) + lat("BLS381", p_bls, 14, (1, 1), (12, 12), beta_bls)
    t2 += "}  // namespace bgls\n"
    out2 = os.path.join(os.path.dirname(OUT), "constants_latx_gen.hpp")
    with open(out2, "w") as f:
        f.write(t2)
    print("wrote", os.path.normpath(out2), len(t2), "bytes")


if __name__ == "__main__":
    main()
