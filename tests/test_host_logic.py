"""CPU-side tests of the product's host logic and boundary (no GPU needed):
the C-ABI library loads and exports every declared symbol, fails loudly without a device, the DER / low-S /
curve gates agree with the oracle and the golden vectors, and the very header code the kernels are compiled
from (run on the CPU through libfabgpu_hosttest.so) is bit-exact against the oracle."""
import ctypes
import json
import os
import re

import numpy as np
import pytest

import bccsp_sw_oracle as po
import coracle
import fabgpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")


def _load(name):
    return json.load(open(os.path.join(G, name)))["vectors"]


def _h32(x):
    return bytes.fromhex(x.rjust(64, "0"))


@pytest.fixture(scope="module")
def hosttest():
    p = os.path.join(ROOT, "fabric-mod_amd", "lib", "libfabgpu_hosttest.so")
    if not os.path.exists(p):
        import __graft_entry__ as g
        g.build()
    return ctypes.CDLL(p)


def test_library_exports_every_declared_symbol():
    L = fabgpu.load()
    declared = set()
    for h in ("fabgpu.h", "fabgpu_bccsp.h"):
        src = open(os.path.join(ROOT, "include", h)).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        declared |= set(re.findall(r"\b(fabgpu_[a-z0-9_]+)\s*\(", src))
    assert declared == set(fabgpu.ABI_SYMBOLS)
    for sym in declared:
        assert hasattr(L, sym), sym
    # ... and NOTHING else named fabgpu_*: the product library's C surface is the two public headers.  Probes, walker
    # comparisons, the synthetic generator and the kernel timer live in libfabgpu_testhooks.so (fabric-mod_amd/csrc/fabgpu_testhooks.h).
    import subprocess
    nm = subprocess.run(["nm", "-D", "--defined-only", fabgpu.lib_path()], capture_output=True, text=True, check=True).stdout
    exported = {ln.split()[-1] for ln in nm.splitlines() if ln.split() and ln.split()[-1].startswith("fabgpu_")}
    assert exported == declared, (sorted(exported - declared), sorted(declared - exported))
    assert not [s_ for s_ in exported if any(w in s_ for w in ("probe", "compare", "synth", "last_kernel", "hosttest"))]
    H = fabgpu.load_hooks()
    nm = subprocess.run(["nm", "-D", "--defined-only", fabgpu.hooks_path()], capture_output=True, text=True, check=True).stdout
    hooks = {ln.split()[-1] for ln in nm.splitlines() if ln.split() and ln.split()[-1].startswith("fabgpu_")}
    assert hooks == set(fabgpu.HOOK_SYMBOLS) and not (hooks & declared)
    hdr = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "fabric-mod_amd", "csrc", "fabgpu_testhooks.h")).read(), flags=re.S)
    assert set(re.findall(r"\b(fabgpu_[a-z0-9_]+)\s*\(", hdr)) == hooks
    for sym in hooks:
        assert hasattr(H, sym), sym
    assert L.fabgpu_abi_version() == 6      # 6: fabgpu_block_pass.ms_stage / device_context, fabgpu_csp_new2 (5: n_device_decoded; 4: pseudonym signatures ride along)
    assert fabgpu.strerror(0) == "ok" and "bccsp/sw" in fabgpu.strerror(-2)


def test_headers_are_plain_c_and_a_c_program_links(tmp_path):
    """The drop-in boundary is a C ABI: tools/abi_smoke.c (C99, no C++) includes both headers, links against libfabgpu.so and,
    without a GPU, reports FABGPU_ENODEV and exits 0 (the caller would keep using bccsp/sw); on an MI355X it verifies the
    RFC 6979 A.2.5 P-256/SHA-256 "sample" signature (low-S mirrored) through the fused entry point."""
    import subprocess
    fabgpu.load()
    exe = str(tmp_path / "abi_smoke")
    libdir = os.path.join(ROOT, "fabric-mod_amd", "lib")
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tools", "abi_smoke.c"),
                    "-L" + libdir, "-lfabgpu", "-o", exe], check=True)
    env = dict(os.environ, LD_LIBRARY_PATH=libdir + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
    out = subprocess.run([exe], env=env, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "verdict=1 status=0" in out.stdout or "bccsp/sw" in out.stdout


def test_no_device_fails_loudly_never_falls_back():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(fabgpu.FabgpuError):
        fabgpu.Context()
    with pytest.raises(fabgpu.FabgpuError):
        fabgpu.GPUCSP()
    # argument validation does not need a device
    L = fabgpu.load()
    assert L.fabgpu_init(None, None) == -1
    assert L.fabgpu_p256_verify_batch(None, 1, None, None, None, None, None, None, None) == -1
    # unknown configuration bits are refused before anything else (fabgpu.h: FABGPU_FLAG_* are 1, 2, 4, ... 256), and so are the two
    # contradictory table flags together
    import ctypes
    for bad in (512, 8 | 16):
        cfg = fabgpu._Cfg(device=-1, max_batch=0, max_arena=0, flags=bad)
        h = ctypes.c_void_p()
        assert L.fabgpu_init(ctypes.byref(cfg), ctypes.byref(h)) == -1 and not h.value
    assert fabgpu.FLAG_ONE_LANE_ONLY | fabgpu.FLAG_TIME_KERNELS | fabgpu.FLAG_NO_QUAD == 7


def test_der_gate_matches_golden_and_oracle():
    for v in _load("der_kats.json"):
        raw = bytes.fromhex(v["der"])
        rc, r, s, fl = fabgpu.unmarshal_ecdsa_signature(raw)
        orc = coracle.der_unmarshal(raw)
        assert (rc != 0) == (not v["ok"]), v["name"]
        assert rc == orc[0], v["name"]
        if v["ok"]:
            assert (r, s, fl) == orc[1:], v["name"]
            assert int.from_bytes(r, "big") == int(v["r"], 16) % (1 << 256)
    # fuzz: mutate good signatures byte-wise, product parser == oracle parser == python restatement
    rng = np.random.default_rng(3)
    good = [bytes.fromhex(v["sig_der"]) for v in _load("ref_cert_kats.json")[:40]]
    for g in good:
        for _ in range(60):
            b = bytearray(g)
            k = rng.integers(0, 4)
            if k == 0:
                b[rng.integers(0, len(b))] = rng.integers(0, 256)
            elif k == 1:
                del b[rng.integers(0, len(b))]
            elif k == 2:
                b.insert(rng.integers(0, len(b)), rng.integers(0, 256))
            else:
                b = b[: rng.integers(0, len(b))]
            raw = bytes(b)
            rc, r, s, fl = fabgpu.unmarshal_ecdsa_signature(raw)
            assert (rc, r, s, fl) == coracle.der_unmarshal(raw) if rc == 0 else rc == coracle.der_unmarshal(raw)[0], raw.hex()
            try:
                R, S = po.unmarshal_ecdsa_signature(raw)
                assert rc == 0 and int.from_bytes(r, "big") == R % (1 << 256)
            except po.BCCSPError:
                assert rc != 0


def test_low_s_curve_and_hash_to_int_gates():
    assert fabgpu.is_low_s(po.HALF_N.to_bytes(32, "big"))
    assert not fabgpu.is_low_s((po.HALF_N + 1).to_bytes(32, "big"))
    assert fabgpu.is_low_s((1).to_bytes(32, "big"))
    for v in _load("edge_kats.json"):
        want = po.on_curve(int(v["qx"], 16), int(v["qy"], 16))
        assert fabgpu.pubkey_on_curve(_h32(v["qx"]), _h32(v["qy"])) == want, v["name"]
    for d in (b"\x01", b"\x00" * 5, bytes(range(32)), bytes(range(40)), bytes(range(64))):
        assert int.from_bytes(fabgpu.hash_to_int(d), "big") == po.hash_to_int(d)


def test_field_arithmetic_headers_against_python_ints(hosttest):
    P, N, R = po.P, po.N, 1 << 256
    Ri, Rni = pow(R, -1, P), pow(R, -1, N)

    def fop(op, a, b=0):
        out = ctypes.create_string_buffer(32)
        hosttest.hosttest_fieldop(op, a.to_bytes(32, "big"), b.to_bytes(32, "big"), out)
        return int.from_bytes(out.raw, "big")
    import random
    rng = random.Random(11)
    special = [0, 1, 2, P - 1, P - 2, 1 << 255, (1 << 224) - 1, (1 << 96) - 1, (1 << 192) + 1, P - (1 << 96)]
    pairs = [(a % P, b % P) for a in special for b in special] + [(rng.randrange(P), rng.randrange(P)) for _ in range(1500)]
    for a, b in pairs:
        assert fop(0, a, b) == a * b * Ri % P
        assert fop(1, a) == a * a * Ri % P
        assert fop(2, a, b) == (a + b) % P
        assert fop(3, a, b) == (a - b) % P
        assert fop(4, a) == a * R % P and fop(5, a) == a * Ri % P
        an, bn = a % N, b % N
        assert fop(6, an, bn) == an * bn * Rni % N and fop(7, an) == an * an * Rni % N
        assert fop(8, an) == an * R % N and fop(9, an) == an * Rni % N
    for a in (P, P + 7, R - 1, N, N + 1):   # to_mont accepts any 256-bit input (off-range keys / digests)
        assert fop(4, a) == a * R % P and fop(8, a) == a * R % N
    for _ in range(10):
        a = rng.randrange(1, N)
        assert fop(10, a * R % N) == pow(a, -1, N) * R % N
        a = rng.randrange(1, P)
        assert fop(11, a * R % P) == pow(a, -1, P) * R % P


def test_generator_comb_table(hosttest):
    for w, d in [(0, 1), (0, 2), (0, 15), (1, 1), (7, 9), (31, 15), (63, 1), (63, 15)]:
        x = ctypes.create_string_buffer(32)
        y = ctypes.create_string_buffer(32)
        hosttest.hosttest_gtab_entry(w, d, x, y)
        assert (int.from_bytes(x.raw, "big"), int.from_bytes(y.raw, "big")) == po.pt_mul(d << (4 * w), (po.GX, po.GY))


def test_fe29_field_arithmetic_against_python_ints(hosttest):
    """fe29.h (the signed 29-bit-limb field the kernels compute in), C bodies, with the accumulator/limb overflow
    assertions of the FE29_CHECK build armed: lazy sums/differences as operands, canonical zero test, round trips."""
    P = po.P

    def op(o, a, b=0):
        out = ctypes.create_string_buffer(32)
        hosttest.hosttest_fe29_op(o, a.to_bytes(32, "big"), b.to_bytes(32, "big"), out)
        return int.from_bytes(out.raw, "big")
    import random
    rng = random.Random(29)
    special = [0, 1, 2, P - 1, P - 2, 1 << 255, (1 << 256) - 1, (1 << 224) - 1, (1 << 96) - 1, 1 << 29, (1 << 29) - 1,
               1 << 28, (1 << 28) - 1, P >> 1, int("1fffffff" * 8, 16), int("10000000" * 8, 16)]
    pairs = [(a, b) for a in special for b in special] + [(rng.randrange(1 << 256), rng.randrange(1 << 256)) for _ in range(1500)]
    for a, b in pairs:
        assert op(0, a, b) == a * b % P
        assert op(1, a) == a * a % P
        assert op(2, a, b) == (a + b) % P
        assert op(3, a, b) == (a - b) % P
        assert op(4, a, b) == (a + b) * (a - b) % P
        assert op(5, a, b) == (1 if (a - b) % P == 0 else 0)
        assert op(6, a) == a % P


def test_safegcd_inversion_against_python_pow(hosttest):
    """modinv30.h: fixed 20 x 30 division steps, mod n (the kernel's w = s^-1) and mod p."""
    import random
    rng = random.Random(30)

    def inv(which, a):
        out = ctypes.create_string_buffer(32)
        hosttest.hosttest_modinv(which, a.to_bytes(32, "big"), out)
        return int.from_bytes(out.raw, "big")
    for which, m in ((0, po.N), (1, po.P)):
        vals = [1, 2, 3, m - 1, m - 2, m >> 1, (m >> 1) + 1, 1 << 255, (1 << 200) + 1, (1 << 30) - 1, 1 << 30] + \
               [1 << k for k in range(0, 256, 17)] + [rng.randrange(1, m) for _ in range(3000)]
        for a in vals:
            assert inv(which, a % m) == pow(a % m, -1, m), (which, hex(a))
        assert inv(which, 0) == 0
    # the pair kernels' lanes compute ONE column of a batch's transition matrix each (modinv_divsteps30_column): same entries, same zeta
    for _ in range(4000):
        zeta = rng.randrange(-600, 2)
        f0 = rng.getrandbits(32) | 1
        assert hosttest.hosttest_divsteps_columns(zeta, f0, rng.getrandbits(32)) == 0


def test_generator_comb_table_fe29(hosttest):
    """16-bit generator comb: T[w][d] = d * 2^(16 w) * G, every window at its corners plus sweeps of windows 0 and 15."""
    cases = [(w, d) for w in range(16) for d in (1, 2, 3, 255, 256, 32767, 32768, 65535)] + [(0, d) for d in range(1, 600)] + \
            [(15, d) for d in range(1, 65536, 977)]
    for w, d in cases:
        x = ctypes.create_string_buffer(32)
        y = ctypes.create_string_buffer(32)
        hosttest.hosttest_gtab29_entry(w, d, x, y)
        assert (int.from_bytes(x.raw, "big"), int.from_bytes(y.raw, "big")) == po.pt_mul(d << (16 * w), (po.GX, po.GY)), (w, d)


def test_combined_mult_adversarial_scalars(hosttest):
    """u1*G + u2*Q through the kernel's CombinedMult for scalar patterns that stress the window recodings: every signed
    5-bit (Booth) digit in {-16..16} at the bottom, middle and top window, runs of ones (carry chains through all 52
    windows), u1 = 0, single bits, n-1, and the exceptional final additions u1*G == +-u2*Q."""
    import random
    rng = random.Random(52)
    N = po.N
    G = (po.GX, po.GY)

    def run(u1, u2, Q):
        x = ctypes.create_string_buffer(32)
        y = ctypes.create_string_buffer(32)
        inf = hosttest.hosttest_combined_mult29(u1.to_bytes(32, "big"), u2.to_bytes(32, "big"), Q[0].to_bytes(32, "big"),
                                                Q[1].to_bytes(32, "big"), x, y)
        return None if inf else (int.from_bytes(x.raw, "big"), int.from_bytes(y.raw, "big"))

    def want(u1, u2, Q):
        return po.pt_add(po.pt_mul(u1, G) if u1 else None, po.pt_mul(u2, Q))
    dq = rng.randrange(1, N)
    Q = po.pt_mul(dq, G)
    u2s = [1, 2, 15, 16, 17, 31, 32, 33, (1 << 255), (1 << 256) % N, N - 1, N - 2, N - 16, N - 17, N >> 1, (1 << 250) - 1,
           int("5" * 64, 16) % N, int("a" * 63, 16), int("f" * 63, 16), int("84210" * 12, 16) % N, int("7bdef" * 12, 16) % N]
    for w in (0, 1, 25, 50, 51):
        for d in range(1, 32):
            v = (d << (5 * w)) % N
            if v:
                u2s.append(v)
    u2s += [rng.randrange(1, N) for _ in range(40)]
    u1s = [0, 1, 255, 256, 65535, 65536, (1 << 240), (1 << 248), N - 1, int("ff00" * 16, 16) % N, int("ffff0000" * 8, 16) % N, int("0001" * 16, 16)] + [rng.randrange(N) for _ in range(8)]
    for k, u2 in enumerate(u2s):
        u1 = u1s[k % len(u1s)]
        assert run(u1, u2, Q) == want(u1, u2, Q), (hex(u1), hex(u2))
    for u1 in u1s:
        assert run(u1, 12345, Q) == want(u1, 12345, Q), hex(u1)
    # exceptional final addition: u1*G == u2*Q (doubling) and u1*G == -u2*Q (infinity): u1 = +-u2*dq
    for u2 in (1, 7, rng.randrange(1, N)):
        assert run(u2 * dq % N, u2, Q) == po.pt_mul(2 * u2 * dq % N, G)
        assert run((-u2 * dq) % N, u2, Q) is None


@pytest.fixture(params=["core29", "core_u256"])
def core_fn(request, hosttest):
    """core29 = the verification core the kernels are compiled from (p256_verify29.h); core_u256 = the saturated-limb
    core kept for the host tools (p256_point.h)."""
    fn = hosttest.hosttest_verify_core29 if request.param == "core29" else hosttest.hosttest_verify_core
    return fn


def _core(core_fn, qx, qy, e, r, s):
    n = qx.shape[0]
    st = np.zeros(n, np.uint8)
    p = lambda a: np.ascontiguousarray(a).ctypes.data_as(ctypes.POINTER(ctypes.c_uint8))
    core_fn(ctypes.c_size_t(n), p(qx), p(qy), p(e), p(r), p(s), p(st))
    return st


def _arr(items):
    return np.frombuffer(b"".join(items), dtype=np.uint8).reshape(-1, 32).copy()


def test_verify_core_headers_on_golden_and_edge_vectors(core_fn):
    vs = [v for v in _load("edge_kats.json") if len(v["e"]) == 64 and 0 <= int(v["r"], 16) < 1 << 256 and 0 <= int(v["s"], 16) < 1 << 256]
    st = _core(core_fn, *[_arr([_h32(v[k]) for v in vs]) for k in ("qx", "qy", "e", "r", "s")])
    for v, got in zip(vs, st):
        assert got == v["status"], v["name"]
    cs = _load("ref_cert_kats.json")
    st = _core(core_fn, *[_arr([_h32(v[k]) for v in cs]) for k in ("qx", "qy", "e", "r", "s")])
    for v, got in zip(cs, st):
        want = po.ST_HIGH_S if not v["low_s"] else (po.ST_VALID if v["expect_valid"] else po.ST_BAD_MATH)
        assert got == want, v["source"]


def test_verify_core_headers_random_vs_oracle(core_fn):
    b = coracle.make_batch(1500, seed=99, invalid_frac=0.25)
    st = _core(core_fn, b["qx"], b["qy"], b["e"], b["r"], b["s"])
    assert (st == coracle.verify_batch(b["qx"], b["qy"], b["e"], b["r"], b["s"])).all()


def test_keyed_core_headers_vs_oracle(hosttest):
    """p256_verify_keyed_core29 (registered public key: both scalar multiplications on comb tables) on the CPU."""
    b = coracle.make_pool_batch(400, seed=31, nkeys=3, invalid_frac=0.25)
    p = lambda a: np.ascontiguousarray(a).ctypes.data_as(ctypes.POINTER(ctypes.c_uint8))
    want = coracle.verify_batch(b["qx"], b["qy"], b["e"], b["r"], b["s"])
    for j in range(3):
        m = b["key_index"] == j
        e, r, s = (np.ascontiguousarray(b[k][m]) for k in ("e", "r", "s"))
        st = np.zeros(int(m.sum()), np.uint8)
        hosttest.hosttest_verify_keyed_core29(ctypes.c_size_t(int(m.sum())), p(b["pool_qx"][j]), p(b["pool_qy"][j]), p(e), p(r), p(s), p(st))
        assert (st == want[m]).all()


def test_synth_generator_is_checked_by_independent_implementations():
    b = fabgpu.synth_batch(800, seed=20260921, invalid_permille=200, threads=4)
    assert (fabgpu.synth_batch(800, seed=20260921, invalid_permille=200, threads=1)["s"] == b["s"]).all()  # thread-count independent
    st = coracle.verify_batch(b["qx"], b["qy"], b["e"], b["r"], b["s"])
    assert (st == coracle.ossl_verify_batch(b["qx"], b["qy"], b["e"], b["r"], b["s"])).all()
    want = np.array([0, 1, 1, 2, 1], dtype=np.uint8)[b["kind"]]
    assert (st == want).all()
    assert (b["kind"] != 0).sum() == 160
    e_in = np.random.default_rng(1).integers(0, 256, (50, 32), dtype=np.uint8)
    c = fabgpu.synth_batch(50, seed=7, e_in=e_in)
    assert (c["e"] == e_in).all() and (coracle.verify_batch(c["qx"], c["qy"], c["e"], c["r"], c["s"]) == 0).all()


def test_coalescer_of_one_signature_calls(hosttest):
    """coalescer.h with a fake device: every caller gets ITS answer, calls that are in flight together share launches, a batch never
    exceeds max_batch, and a failed launch is a failure for exactly the callers it carried."""
    import ctypes
    f = hosttest.hosttest_coalescer
    f.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_int] + [ctypes.POINTER(ctypes.c_uint64)] * 3
    f.restype = ctypes.c_int
    launches, largest, failed = ctypes.c_uint64(0), ctypes.c_uint64(0), ctypes.c_uint64(0)
    # 24 callers x 40 calls, a launch takes 300 us: callers pile up behind the running launch
    assert f(24, 40, 300, 50, 32768, 0, ctypes.byref(launches), ctypes.byref(largest), ctypes.byref(failed)) == 0
    assert launches.value < 24 * 40 / 4 and largest.value >= 8 and failed.value == 0
    # one caller alone: one launch per call (the window only delays it)
    assert f(1, 20, 0, 20, 32768, 0, ctypes.byref(launches), ctypes.byref(largest), ctypes.byref(failed)) == 0
    assert launches.value == 20 and largest.value == 1
    # max_batch 4: no launch carries more
    assert f(16, 25, 200, 0, 4, 0, ctypes.byref(launches), ctypes.byref(largest), ctypes.byref(failed)) == 0
    assert largest.value <= 4 and launches.value >= 16 * 25 / 4
    # every third launch fails: its callers see the failure, nobody sees a wrong answer
    assert f(12, 30, 100, 0, 32768, 3, ctypes.byref(launches), ctypes.byref(largest), ctypes.byref(failed)) == 0
    assert 0 < failed.value < 12 * 30


def test_pass_routing_spreads_blocks_over_the_pool(hosttest):
    """pass_route.h (GPUCSP::RouteBlock): the device with the fewest passes in flight, ties broken round the ring from block_seq mod G.
    The reference has ONE process-global BCCSP (bccsp/factory/factory.go:41-55) whose callers - the channels of a peer - arrive side by
    side (core/committer/txvalidator/v20/validator.go:194-210): the provider, not its callers, spreads them over the node's GPUs."""
    route = hosttest.hosttest_route_block
    route.argtypes = [ctypes.c_uint64, ctypes.POINTER(ctypes.c_uint32), ctypes.c_int]

    def r(seq, fl):
        return route(seq, (ctypes.c_uint32 * len(fl))(*fl), len(fl))
    assert r(12345, [7]) == 0                                                  # one device: always it
    for G in (2, 3, 8):
        assert [r(s, [0] * G) for s in range(2 * G)] == [s % G for s in range(2 * G)]           # idle pool: round the ring by name
        for busy in range(G):                                                  # one idle device among busy ones always wins
            fl = [2] * G
            fl[busy] = 0
            assert all(r(s, fl) == busy for s in range(3 * G))
        # ties among the least busy: the first one round the ring from seq mod G
        fl = [1, 0] + [1] * (G - 2) if G > 2 else [1, 1]
        if G > 2:
            fl[-1] = 0
            assert r(2, fl) == G - 1 and r(0, fl) == 1 and r(1, fl) == 1
    # a long run of arrivals that each stay in flight: every device ends up with the same load
    G, fl = 8, [0] * 8
    rng = np.random.default_rng(5)
    for _ in range(800):
        fl[r(int(rng.integers(0, 1 << 62)), fl)] += 1
    assert fl == [100] * 8
    # passes come and go (each arrival finds the previous arrivals' devices busy, a random one finishes): never more than one apart
    fl = [0] * 8
    for k in range(4000):
        fl[r(int(rng.integers(0, 1 << 62)), fl)] += 1
        if k >= 8:
            busy = [g for g in range(8) if fl[g]]
            fl[busy[int(rng.integers(0, len(busy)))]] -= 1
        assert max(fl) - min(fl) <= 2, fl


def test_certificate_walk_over_a_window_of_the_der(hosttest):
    """The device's certificate decoder keeps the first 3 KiB of a certificate's DER and walks it with the real length in hand
    (block_walk_core.h cert_der_p256_key_offset_window).  Over every certificate fixture of the reference and every window size: the
    answer is the whole-certificate answer, or -2 ("needs bytes beyond the window") - never a different key offset, never a wrong
    "no key"; and once the window reaches the end of SubjectPublicKeyInfo it IS the whole-certificate answer."""
    import base64
    fn = hosttest.hosttest_cert_key_offset_window
    fn.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_size_t]
    chains = json.load(open(os.path.join(G, "ref_cert_chains.json")))
    ders = [base64.b64decode(v) for _, v in sorted(chains["certs"].items())]
    n_key = n_not = 0
    for der in ders:
        full = fn(der, len(der), len(der))
        assert full >= -1
        n_key += full >= 0
        n_not += full < 0
        padded = der + bytes(5000)                                             # bytes behind the Certificate change nothing
        assert fn(padded, len(padded), len(padded)) == full
        seen_decided = False
        for avail in list(range(0, min(len(der), 700))) + [len(der) - 1, len(der)]:
            got = fn(der, avail, len(der))
            assert got in (full, -2), (avail, got, full)
            if full >= 0:
                assert (got == full) == (avail >= full + 64), (avail, got, full)   # decided exactly when X || Y is inside the window
            if got != -2:
                seen_decided = True
            elif seen_decided:
                raise AssertionError("a larger window must not take a decided answer back")
        # truncated certificates (len < the real length): a length that does not cover the encoding is "no key", window or not
        for cut in (1, 10, len(der) // 2):
            assert fn(der, len(der) - cut, len(der) - cut) == -1
    assert n_key > 60, (n_key, n_not)
    # certificates WITHOUT a P-256 key (another curve's OID, a broken BIT STRING header): "no key" or "needs more", never a key
    for der in ders[:20]:
        at = fn(der, len(der), len(der))
        for delta, val in ((-4, 0x22), (-1, 0x05), (-2, 0x01), (-11, 0x2B)):   # inside prime256v1's OID / the 00 04 prefix / the lengths in front
            m = bytearray(der)
            m[at + delta] = val
            full = fn(bytes(m), len(m), len(m))
            assert full == -1, (delta, full)
            for avail in range(0, at + 80, 7):
                assert fn(bytes(m), avail, len(m)) in (-1, -2)
            n_not += 1
    assert n_not >= 80


# ---- bench.py's compact line (a 23 KB line came back from the driver as `parsed: null`) ----
def _bench_module():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_bench_compact_line_from_round4_detail():
    """the full object round 4 printed (23 KB, kept as profiles/r04_bench_final.json) -> one line under 4 KB with every contract key"""
    import json
    bench = _bench_module()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    full = json.load(open(os.path.join(root, "profiles", "r04_bench_final.json")))
    assert len(json.dumps(full)) > 20000
    text = bench.compact_line(full)
    assert "\n" not in text and len(text) < 4096, len(text)
    line = json.loads(text, parse_constant=lambda c: (_ for _ in ()).throw(ValueError("non-strict JSON constant " + c)))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "cpu_baseline", "parity"):
        assert k in line, k
    assert line["config"]["workload"].startswith("BASELINE.json configs[1]")
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "kernel_ms"):
        assert k in line["roofline"], k
    for k in ("value", "unit", "cores", "kind", "cpu_model"):
        assert k in line["cpu_baseline"], k
    assert abs(line["value"] - full["value"]) / full["value"] < 1e-5
    assert abs(line["roofline"]["frac"] - full["roofline"]["frac"]) < 1e-5
    assert line["validated_tx_per_s_block_pass"] > 1e6 and line["idemix_roofline_frac"] < 1
    assert all(not isinstance(v, (dict, list)) or k in ("config", "roofline", "cpu_baseline", "fresh_provider_lone_passes_ms", "legs_with_errors") for k, v in line.items())


def test_bench_compact_line_worst_case_stays_under_the_limit():
    """every optional leg present, long strings everywhere, errors in legs: still < 4 KB, still strict JSON, contract keys never shed"""
    import json
    bench = _bench_module()
    long = "x" * 5000
    out = {"metric": "ECDSA P-256 verifies/sec (whole node)", "value": 1.23456789e8, "unit": "verifies/s", "n_gpus": 8, "steps": 20, "warmup": 5,
           "ms_per_step": 0.7123456789, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int64",
           "data": "synthetic - DRY RUN of the multi-rank code path" + long,
           "config": {"workload": long, "tuples_per_gpu": 30000, "tx_per_block": 10000, "endorsements_per_tx": 3, "seed": 20260921, "parallelism": "1 block per GPU"},
           "roofline": {"bound": "valu-mac", "achieved": 1.3e13, "peak": 3.4e13, "unit": "MAC/s", "frac": 0.4, "traffic": 9.6e7, "traffic_source": "profiles/x.json",
                        "kernel": "k<256> (" + long + ")", "kernel_ms": 0.68, "hbm": {"achieved": 7.0, "frac": 0.0009}, "model": long},
           "cpu_baseline": {"value": 4.6e5, "unit": "verifies/s", "cores": 16, "kind": "port", "cpu_model": "AMD EPYC 9575F 64-Core Processor", "single_thread": {"value": 2.9e4},
                            "thread_sweep": {str(i): {"best": 1.0, "median": 1.0} for i in range(300)}},
           "parity": long, "configs2_strong": {"value": 4.0e7}, "configs2_inprocess": {"error": long}, "configs3_fused": {"value": 5.3e7},
           "configs4_mixed": {"value": 2.6e7, "ms_per_step": 1.1, "roofline": {"frac": 0.083, "kernel_ms": 0.77}, "mixed_step_over_the_longer_kernel": 1.45},
           "block_pass": {"error": long}, "block_pass_inprocess": {"error": long}, "rccl_ranks": 8,
           "validated_tx_per_s": 1.4e7, "value_pcie_inclusive": 3.5e7}
    text = bench.compact_line(out, "bench_detail_n8.json")
    assert len(text) < 4096
    line = json.loads(text)
    assert line["n_gpus"] == 8 and line["data"].startswith("synthetic (DRY RUN")
    assert set(line["legs_with_errors"]) == {"configs2_inprocess", "block_pass", "block_pass_inprocess"}
    assert line["roofline"]["kernel"] == "k<256>" and line["detail"] == "bench_detail_n8.json"
    # NaN / inf never reach the line as bare tokens a strict parser rejects
    out["value_pcie_inclusive"] = float("nan")
    out["roofline"]["traffic"] = float("inf")
    again = bench.compact_line(out)
    assert "NaN" not in again and "Infinity" not in again and json.loads(again)["roofline"]["traffic"] is None


# ---- idemix issuer keys: only the encoding golang/protobuf itself produces is accelerated (idemix/issuerkey.go:171-182) ----
def _pb_fields(raw):
    out, p = [], 0
    while p < len(raw):
        start = p
        tag = raw[p]
        p += 1
        ln, k = 0, 0
        while True:
            c = raw[p]
            p += 1
            ln |= (c & 0x7F) << (7 * k)
            k += 1
            if not c & 0x80:
                break
        out.append((tag >> 3, raw[start:p + ln]))
        p += ln
    return out


def test_issuer_key_canonical_encoding_gate():
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    L = fabgpu.load()
    canon = lambda b: L.fabgpu_idemix_issuer_key_is_canonical(bytes(b), len(b))
    fx = json.load(open(os.path.join(root, "tests", "golden", "idemix_fixtures.json")))["msps"]
    for name, m in fx.items():
        raw = bytes.fromhex(m["ipk"])
        assert canon(raw) == 1, name                                         # the reference's own testdata keys (idemixgen output)
        f = _pb_fields(raw)
        assert b"".join(x for _, x in f) == raw
        # the same fields in another order: proto.Unmarshal accepts it, proto.Marshal would emit other bytes -> SetHash hashes other bytes
        swapped = list(f)
        i2, i3 = [k for k, (n, _) in enumerate(f) if n == 2][0], [k for k, (n, _) in enumerate(f) if n == 3][0]
        swapped[i2], swapped[i3] = swapped[i3], swapped[i2]
        assert canon(b"".join(x for _, x in swapped)) == 0
        # a duplicate singular field (last one wins on unmarshal, one copy is marshalled)
        assert canon(raw + [x for n, x in f if n == 10][0]) == 0
        # an unknown field (golang/protobuf keeps it in XXX_unrecognized and emits it at the END, wherever it stood) / a varint field
        assert canon(raw[:4] + bytes([0x5A, 0x01, 0x00]) + raw[4:]) == 0     # field 11 in the middle
        assert canon(raw + bytes([0x5A, 0x01, 0x00])) == 0                   # ... and at the end: not a field of idemix.proto:37-48 at all
        assert canon(raw + bytes([0x58, 0x01])) == 0                         # field 11 as a varint
        # a padded length varint (0xA0 0x00 for 32)
        h = [x for n, x in f if n == 10][0]
        padded = raw[:len(raw) - len(h)] + bytes([0x52, 0xA0, 0x00]) + h[2:]
        assert _pb_fields(padded)[-1][0] == 10 and canon(padded) == 0
        # an empty singular bytes field (proto3 does not marshal it) / an empty repeated string (it does)
        assert canon(raw[:len(raw) - len(h)] + bytes([0x52, 0x00])) == 0
        assert canon(bytes([0x0A, 0x00]) + raw) == 1
        # a nested ECP with its coordinates the wrong way round
        hsk = f[i2][1]
        inner = _pb_fields(hsk[2:])
        bad_hsk = hsk[:2] + inner[1][1] + inner[0][1]
        assert canon(b"".join(bad_hsk if k == i2 else x for k, (_, x) in enumerate(f))) == 0
    assert canon(b"") == 0 and L.fabgpu_idemix_issuer_key_is_canonical(None, 0) == 0


# ---- every kernel of the product's translation units is resolved at provider construction (round 5: warm_kernel_functions_*) ----
def test_every_global_kernel_is_in_its_units_warm_list():
    """GPUCSP::Preallocate asks for the attributes of every __global__ function so that none is resolved at its first launch inside a
    block's pass (DESIGN.md 4.4d "Round 5").  A kernel added to a unit must be added to that unit's list: this test reads the sources."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    csrc = os.path.join(root, "fabric-mod_amd", "csrc")
    for unit, fn in (("kernels.hip", "warm_kernel_functions_kernels"), ("wide_kernels.hip", "warm_kernel_functions_wide"),
                     ("idemix_kernels.hip", "warm_kernel_functions_idemix"), ("block_walk_kernels.hip", "warm_kernel_functions_walk"),
                     ("keytab_kernels.hip", "warm_kernel_functions_keytab")):
        src = open(os.path.join(csrc, unit)).read()
        kernels = set(re.findall(r"__global__\s+void\s+(?:__launch_bounds__\([^)]*\)\s*)?(\w+)\s*\(", src))
        assert len(kernels) >= (3 if unit.startswith("keytab") else 5), (unit, kernels)
        body = src[src.index("int %s()" % fn):]
        body = body[:body.index("return ok;")]
        listed = set(re.findall(r"\(const void\*\)\s*\(?\s*(\w+)", body))
        assert kernels <= listed, "%s: kernels missing from %s: %s" % (unit, fn, sorted(kernels - listed))
    api = open(os.path.join(csrc, "fabgpu_api.hip")).read()
    for fn in ("warm_kernel_functions_kernels", "warm_kernel_functions_wide", "warm_kernel_functions_idemix", "warm_kernel_functions_walk",
               "warm_kernel_functions_keytab"):
        assert "(void)%s();" % fn in api
