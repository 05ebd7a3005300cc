"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C ABI, against the CPU
oracle and the committed golden fixtures.  Integer/byte work: everything is compared BIT-EXACT.
Structure follows the reference's tests for this path:
  bccsp/sw/impl_test.go:525-586 TestECDSAVerify, :968-1033 TestECDSALowS, :928-966 TestECDSASignatureEncoding,
  :1293-1339 TestSHA, bccsp/sw/sw_test.go:130-149 (argument errors), msp/msp_test.go:494-536 (sign/verify/tamper),
  core/common/validation/fullflow_test.go:240-250 (every single-byte mutation flips the verdict)."""
import hashlib
import json
import os

import numpy as np
import pytest

import bccsp_sw_oracle as po
import coracle
import fabgpu

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")


def _load(name):
    return json.load(open(os.path.join(G, name)))["vectors"]


def _h32(x):
    return bytes.fromhex(x.rjust(64, "0"))


def _arr(items):
    return np.frombuffer(b"".join(items), dtype=np.uint8).reshape(-1, 32).copy()


@pytest.fixture(scope="module", params=["auto", "one-lane", "pair-table-lds", "pair-table-global", "no-wide"])
def ctx(request):
    """auto = the product default (two lanes per signature up to 32 768 tuples, one lane beyond; registered keys: eight lanes per
    signature in two phases up to 8 192 signatures - p256_wide29.h); no-wide = FABGPU_FLAG_NO_WIDE: registered keys on the two-lane /
    one-lane keyed kernels at every size, so that every keyed fixture runs through both forms;
    one-lane = FABGPU_FLAG_ONE_LANE_ONLY, so that every fixture and edge vector also goes through the one-lane kernel;
    pair-table-lds / -global = the two homes of the pair kernel's per-signature table (8 entries + signed 4-bit windows in LDS / 16
    entries + 5-bit windows in the global workspace): every fixture and edge vector through both, whichever is the default."""
    extra = {"auto": 0, "one-lane": fabgpu.FLAG_ONE_LANE_ONLY, "pair-table-lds": fabgpu.FLAG_PAIR_TABLE_LDS, "pair-table-global": fabgpu.FLAG_PAIR_TABLE_GLOBAL,
             "no-wide": fabgpu.FLAG_NO_WIDE}
    c = fabgpu.Context(device=0, flags=fabgpu.FLAG_TIME_KERNELS | extra[request.param])
    yield c
    c.close()


def test_scalars_at_the_window_recodings_edges(ctx):
    """VALID signatures whose u2 = r / s (the scalar that multiplies the public key) sits where a signed-window recoding can go wrong:
    n - 2 (with 4-bit windows the ONE scalar whose last addition is a doubling, T = -Q plus -Q: p256_pair29.h takes -2Q from the table
    instead), n - 1, small values, all-ones digits, the values around every power of 16 and 32 at the top.  A signer can hit any u2 it
    likes by choosing its private key after the fact: for a digest e and a nonce k, r = x(kG), s = r / u2, d = (s k - e) / r."""
    rng = np.random.default_rng(4)
    targets = [po.N - 2, po.N - 1, po.N - 3, po.N - 16, po.N - 17, po.N - 18, po.N - 32, po.N - 34, 1, 2, 3, 7, 8, 9, 15, 16, 17, 31, 32, 33,
               (1 << 255) - 1, 1 << 255, (1 << 255) + 8, (1 << 256) % po.N, int("8" * 64, 16) % po.N, int("7" * 64, 16), int("f" * 63, 16),
               int("1" * 64, 16), po.N >> 1, (po.N >> 1) + 1]
    rows, want = [], []
    for u2 in targets:
        for _ in range(200):
            k = int(rng.integers(1, 1 << 62)) * int(rng.integers(1, 1 << 62)) + 1
            R = po.pt_mul(k, (po.GX, po.GY))
            r = R[0] % po.N
            s = r * pow(u2, -1, po.N) % po.N
            if r == 0 or not po.is_low_s(s):
                continue
            digest = bytes(rng.integers(0, 256, size=32, dtype=np.uint8))
            e = int.from_bytes(digest, "big")
            d = (s * k - e) * pow(r, -1, po.N) % po.N
            if d == 0:
                continue
            Q = po.pt_mul(d, (po.GX, po.GY))
            assert po.ecdsa_verify_raw(Q[0], Q[1], digest, r, s)
            rows.append((_h32("%064x" % Q[0]), _h32("%064x" % Q[1]), digest, _h32("%064x" % r), _h32("%064x" % s)))
            want.append(0)
            rows.append((_h32("%064x" % Q[0]), _h32("%064x" % Q[1]), digest, _h32("%064x" % r), _h32("%064x" % ((s + 1) % po.N or 1))))   # same key, another s
            want.append(None)
            break
        else:
            raise AssertionError("no low-S signature found for u2 = %x" % u2)
    cols = [_arr([row[c] for row in rows]) for c in range(5)]
    bits, st = ctx.p256_verify_batch(*cols)
    oracle = coracle.verify_batch(*cols)
    assert (st == oracle).all()
    for i, w in enumerate(want):
        if w is not None:
            assert st[i] == 0 and bits[i], (i, hex(targets[i // 2]))


@pytest.fixture(scope="module")
def csp():
    c = fabgpu.GPUCSP(device=0)
    yield c
    c.close()


def test_native_library_is_the_thing_running(ctx):
    # the HIP extension is loaded in-tree and bound to a gfx950 device; there is no other code path
    assert os.path.exists(fabgpu.lib_path())
    assert ctx.device_count() >= 1
    with open("/proc/self/maps") as f:
        assert "libfabgpu.so" in f.read()


def test_plain_c_program_verifies_the_rfc6979_vector_through_the_c_abi(tmp_path):
    import subprocess
    exe = str(tmp_path / "abi_smoke")
    libdir = os.path.dirname(fabgpu.lib_path())
    subprocess.run(["gcc", "-std=c99", "-Wall", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tools", "abi_smoke.c"),
                    "-L" + libdir, "-lfabgpu", "-o", exe], check=True)
    env = dict(os.environ, LD_LIBRARY_PATH=libdir + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
    out = subprocess.run([exe], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "verdict=1 status=0" in out.stdout, out.stdout + out.stderr


# ---- golden fixtures -----------------------------------------------------------------------------
def test_reference_cert_fixture_kats(ctx):
    vs = _load("ref_cert_kats.json")
    bits, st = ctx.p256_verify_batch(*[_arr([_h32(v[k]) for v in vs]) for k in ("qx", "qy", "e", "r", "s")])
    for v, b, s in zip(vs, bits, st):
        want = po.ST_HIGH_S if not v["low_s"] else (po.ST_VALID if v["expect_valid"] else po.ST_BAD_MATH)
        assert s == want and b == (want == 0), v["source"]
    assert sum(1 for v in vs if v["pinned_by"]) >= 60


def test_edge_vectors(ctx):
    vs = [v for v in _load("edge_kats.json") if 0 <= int(v["r"], 16) < 1 << 256 and 0 <= int(v["s"], 16) < 1 << 256]
    e = [fabgpu.hash_to_int(bytes.fromhex(v["e"])) for v in vs]     # hashToInt on the host, as the Go provider does
    bits, st = ctx.p256_verify_batch(_arr([_h32(v["qx"]) for v in vs]), _arr([_h32(v["qy"]) for v in vs]), _arr(e),
                                     _arr([_h32(v["r"]) for v in vs]), _arr([_h32(v["s"]) for v in vs]))
    for v, b, s in zip(vs, bits, st):
        assert s == v["status"] and b == (v["status"] == 0), v["name"]


def test_rfc6979_public_vectors(ctx):
    f = json.load(open(os.path.join(G, "rfc6979_p256_sha256.json")))
    qx, qy = _h32(f["qx"]), _h32(f["qy"])
    rows = []
    for v in f["vectors"]:
        s = int(v["s"], 16)
        rows.append((v["message"].encode(), _h32(v["r"]), _h32(v["s"]), po.ST_VALID if po.is_low_s(s) else po.ST_HIGH_S))
        rows.append((v["message"].encode(), _h32(v["r"]), (po.N - s).to_bytes(32, "big"), po.ST_VALID if po.is_low_s(po.N - s) else po.ST_HIGH_S))
    msgs = [m for m, _, _, _ in rows]
    off = np.concatenate([[0], np.cumsum([len(m) for m in msgs])]).astype(np.uint32)
    arena = np.frombuffer(b"".join(msgs) + b"\0", dtype=np.uint8)
    n = len(rows)
    bits, st = ctx.sha256_p256_verify_batch(arena, off, _arr([qx] * n), _arr([qy] * n), _arr([r for _, r, _, _ in rows]), _arr([s for _, _, s, _ in rows]))
    assert list(st) == [w for _, _, _, w in rows] and list(bits) == [w == 0 for _, _, _, w in rows]
    kid = ctx.key_register(qx, qy)
    bits2, st2 = ctx.sha256_p256_verify_batch_keyed(arena, off, np.full(n, kid, dtype=np.uint32), _arr([r for _, r, _, _ in rows]), _arr([s for _, _, s, _ in rows]))
    assert (st2 == st).all()


# ---- seeded random batches vs oracle, ragged sizes ---------------------------------------------------
@pytest.mark.parametrize("n", [1, 2, 31, 32, 33, 63, 64, 65, 127, 128, 129, 255, 256, 257, 1000, 4097, 32767, 32768, 32769])
def test_verify_batch_sizes(ctx, n):
    b = coracle.make_batch(n, seed=1000 + n, invalid_frac=0.2 if n > 4 else 0.0)
    bits, st = ctx.p256_verify_batch(b["qx"], b["qy"], b["e"], b["r"], b["s"])
    want = coracle.verify_batch(b["qx"], b["qy"], b["e"], b["r"], b["s"])
    assert (st == want).all() and (bits == (want == 0)).all()
    bits2, none = ctx.p256_verify_batch(b["qx"], b["qy"], b["e"], b["r"], b["s"], want_status=False)
    assert none is None and (bits2 == bits).all()


def test_empty_batch(ctx):
    z = np.zeros((0, 32), np.uint8)
    bits, st = ctx.p256_verify_batch(z, z, z, z, z)
    assert bits.size == 0 and st.size == 0
    assert ctx.sha256_batch(np.zeros(1, np.uint8), np.zeros(1, np.uint32)).shape == (0, 32)


def test_baseline_cfg2_block_10k_tx_x3(ctx):
    # BASELINE.json configs[1]: 10k tx x 3 endorsements, 1 % invalid mix; product generator, oracle verdicts
    n = 30000
    b = fabgpu.synth_batch(n, seed=20260921, invalid_permille=10)
    bits, st = ctx.p256_verify_batch(b["qx"], b["qy"], b["e"], b["r"], b["s"])
    want = coracle.verify_batch(b["qx"], b["qy"], b["e"], b["r"], b["s"])
    assert (st == want).all() and (bits == (want == 0)).all()
    assert (want == np.array([0, 1, 1, 2, 1], dtype=np.uint8)[b["kind"]]).all()
    tx_ok = bits.reshape(10000, 3).all(axis=1)                     # a tx is valid iff all its endorsements verify
    assert tx_ok.sum() == 10000 - len(set(np.nonzero(b["kind"])[0] // 3))


def test_full_size_cfg4_properties(ctx):
    # BASELINE.json configs[3] size (100k tx x 3): size-independent properties instead of a per-item oracle run:
    # verdict == (no mutation applied), a checksum of the verdict words, and idempotence of a second launch.
    n = 300000
    b = fabgpu.synth_batch(n, seed=4, invalid_permille=10)
    bits, st = ctx.p256_verify_batch(b["qx"], b["qy"], b["e"], b["r"], b["s"])
    assert (bits == (b["kind"] == 0)).all()
    assert (st == np.array([0, 1, 1, 2, 1], dtype=np.uint8)[b["kind"]]).all()
    bits2, st2 = ctx.p256_verify_batch(b["qx"], b["qy"], b["e"], b["r"], b["s"])
    assert (bits2 == bits).all() and (st2 == st).all()
    sample = np.random.default_rng(0).choice(n, 3000, replace=False)   # oracle on a sample
    want = coracle.verify_batch(b["qx"][sample], b["qy"][sample], b["e"][sample], b["r"][sample], b["s"][sample])
    assert (st[sample] == want).all()


def test_full_size_cfg4_fused_kernel_300k(ctx):
    """BASELINE.json configs[3] through the kernel it names: 100k tx x 3 = 300 000 messages of 1 856 B (prp 1024 shared by a tx's three
    endorsements + endorser 832), SHA-256 FUSED ahead of the verify in one launch.  Every verdict bit against the generator's ground
    truth (kind vector), 12 000 tuples - every mutated one among them - status-exact against the C oracle's own SHA-256 + verify,
    idempotence, and the digests the fused kernel hands out against hashlib."""
    import hashlib
    n_tx, n, L = 100000, 300000, 1856
    rng = np.random.default_rng(20260921)
    arena = np.empty((n, L), dtype=np.uint8)
    arena[:, :1024] = np.repeat(rng.integers(0, 256, size=(n_tx, 1024), dtype=np.uint8), 3, axis=0)
    arena[:, 1024:] = rng.integers(0, 256, size=(n, L - 1024), dtype=np.uint8)
    off = (np.arange(n + 1, dtype=np.uint64) * L).astype(np.uint32)
    flat = arena.reshape(-1)
    dig = ctx.sha256_batch(flat, off)
    b = fabgpu.synth_batch(n, seed=5, invalid_permille=10, e_in=dig)
    bits, st = ctx.sha256_p256_verify_batch(flat, off, b["qx"], b["qy"], b["r"], b["s"])
    # kind 1 mutates e_out only: in hash mode the message decides, so those tuples stay valid
    want_kind = np.array([0, 0, 1, 2, 1], dtype=np.uint8)[b["kind"]]
    assert (st == want_kind).all() and (bits == (want_kind == 0)).all()
    bad = np.nonzero(b["kind"] > 1)[0]
    samp = np.unique(np.concatenate([bad, rng.choice(n, size=10000, replace=False)]))[:12000]
    soff = (np.arange(len(samp) + 1, dtype=np.uint64) * L).astype(np.uint32)
    want = coracle.sha256_verify_batch(arena[samp].reshape(-1), soff, b["qx"][samp], b["qy"][samp], b["r"][samp], b["s"][samp])
    assert (st[samp] == want).all() and len(bad) > 1500
    for i in rng.choice(n, size=500, replace=False):
        assert dig[i].tobytes() == hashlib.sha256(arena[i].tobytes()).digest()
    bits2, st2 = ctx.sha256_p256_verify_batch(flat, off, b["qx"], b["qy"], b["r"], b["s"])
    assert (bits2 == bits).all() and (st2 == st).all()
    # the same block with the prp handed over once per transaction (shared prefix, mid-state reuse) and digests handed out
    pre = np.ascontiguousarray(arena[::3, :1024]).reshape(-1)
    suf = np.ascontiguousarray(arena[:, 1024:]).reshape(-1)
    flat2 = np.concatenate([pre, suf])
    pre_off = (np.arange(n_tx + 1, dtype=np.uint64) * 1024).astype(np.uint32)
    off2 = (n_tx * 1024 + np.arange(n + 1, dtype=np.uint64) * (L - 1024)).astype(np.uint32)
    r3 = ctx.identity_verify_batch(flat2, off2, b["r"], b["s"], qx=b["qx"], qy=b["qy"], pre_off=pre_off, pre_idx=np.repeat(np.arange(n_tx, dtype=np.uint32), 3),
                                   want_digests=True)
    assert (r3[1] == st).all() and (r3[0] == bits).all()
    assert (r3[2] == dig).all()                                       # 300 000 digests out of the fused kernel == the plain SHA-256 kernel's


def test_arena_size_boundaries_of_the_32_bit_offsets(ctx):
    """off[] is u32: the C ABI must refuse what it cannot address (FABGPU_ETOOBIG, an infrastructure error -> bccsp/sw), never wrap."""
    import ctypes
    L = fabgpu.load()
    n = 4
    z = np.zeros((n, 32), np.uint8)
    words = np.zeros(1, np.uint64)
    u8 = ctypes.POINTER(ctypes.c_uint8)
    u32 = ctypes.POINTER(ctypes.c_uint32)
    # device entry point: an arena of 2^32 bytes and more is refused before anything is launched (no such allocation is touched)
    for fn, args in ((L.fabgpu_sha256_batch_dev, (ctx.handle, n, 1 << 20, (1 << 32), 1 << 20, 1 << 20, None)),
                     (L.fabgpu_sha256_p256_verify_batch_dev, (ctx.handle, n, 1 << 20, (1 << 32) + 5, 1 << 20, 1 << 20, 1 << 20, 1 << 20, 1 << 20, 1 << 20, None, None))):
        assert fn(*args) == -5
    # host entry point: descending offsets are inconsistent arguments, a tuple count beyond the staging limit is too big
    off = np.array([0, 10, 5, 20, 30], dtype=np.uint32)
    assert L.fabgpu_sha256_batch(ctx.handle, n, np.zeros(64, np.uint8).ctypes.data_as(u8), off.ctypes.data_as(u32), np.zeros((n, 32), np.uint8).ctypes.data_as(u8)) == -1
    assert L.fabgpu_p256_verify_batch(ctx.handle, (0x7FFFFFF0 // 160) + 1, z.ctypes.data_as(u8), z.ctypes.data_as(u8), z.ctypes.data_as(u8), z.ctypes.data_as(u8),
                                      z.ctypes.data_as(u8), words.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)), None) == -5
    # a large but legal arena: 1.2 GB of messages in one launch (2 000 messages of 600 000 bytes: 9 375 blocks per lane)
    m, ml = 2000, 600000
    big = np.random.default_rng(9).integers(0, 256, size=m * ml, dtype=np.uint8)
    boff = (np.arange(m + 1, dtype=np.uint64) * ml).astype(np.uint32)
    d = ctx.sha256_batch(big, boff)
    import hashlib
    for i in (0, 1, 777, m - 1):
        assert d[i].tobytes() == hashlib.sha256(big[i * ml:(i + 1) * ml].tobytes()).digest()


# ---- registered public keys (bccsp.KeyImport -> per-key comb table) --------------------------------------
@pytest.mark.parametrize("n,nkeys", [(1, 1), (33, 2), (500, 16), (4097, 16), (33000, 16)])
def test_registered_keys_pool_vs_oracle_and_vs_the_unkeyed_kernel(ctx, n, nkeys):
    b = coracle.make_pool_batch(n, seed=4242 + n, nkeys=nkeys, invalid_frac=0.2 if n > 4 else 0.0)
    ids = np.array([ctx.key_register(b["pool_qx"][j].tobytes(), b["pool_qy"][j].tobytes()) for j in range(nkeys)], dtype=np.uint32)
    assert len(set(ids.tolist())) == nkeys
    assert ctx.key_register(b["pool_qx"][0].tobytes(), b["pool_qy"][0].tobytes()) == ids[0]        # idempotent
    bits, st = ctx.p256_verify_batch_keyed(ids[b["key_index"]], b["e"], b["r"], b["s"])
    want = coracle.verify_batch(b["qx"], b["qy"], b["e"], b["r"], b["s"])
    assert (st == want).all() and (bits == (want == 0)).all()
    bits2, st2 = ctx.p256_verify_batch(b["qx"], b["qy"], b["e"], b["r"], b["s"])
    assert (st2 == st).all() and (bits2 == bits).all()


def test_registered_keys_with_16_bit_tables():
    """Round 6, FABGPU_FLAG_KEY_TABLES_16BIT: every registered key also gets a 16-bit comb (80 MiB, built behind the registration); a
    wavefront all of whose keys have one takes 16 mixed additions for u2 Q instead of 32.  (1) The statuses are the oracle's BEFORE the
    tables are there (8-bit path), AFTER (16-bit path) and for a batch that mixes keys with and without one (the 65th key gets none:
    wavefronts that meet it fall back) - in both geometries, verify-only and fused.  (2) Each 16-bit table agrees with the key's 8-bit
    table entry for entry where the two overlap (the test hook reads both from the device)."""
    ctx = fabgpu.Context(device=0, flags=fabgpu.FLAG_KEY_TABLES_16BIT)
    try:
        nkeys = 6
        b = coracle.make_pool_batch(40000, seed=9191, nkeys=nkeys, invalid_frac=0.1)
        ids = np.array([ctx.key_register(b["pool_qx"][j].tobytes(), b["pool_qy"][j].tobytes()) for j in range(nkeys)], dtype=np.uint32)
        want = coracle.verify_batch(b["qx"], b["qy"], b["e"], b["r"], b["s"])
        for n in (3000, 40000):                                                      # pair and one-lane geometry, tables possibly still building
            bits, st = ctx.p256_verify_batch_keyed(ids[b["key_index"][:n]], b["e"][:n], b["r"][:n], b["s"][:n])
            assert (st == want[:n]).all() and (bits == (want[:n] == 0)).all()
        for j in range(nkeys):
            assert ctx.test_key_tables16(int(ids[j])) == nkeys                        # waits for the builds; cross-checks table j
        for n in (1, 77, 3000, 9000, 30000, 40000):
            bits, st = ctx.p256_verify_batch_keyed(ids[b["key_index"][:n]], b["e"][:n], b["r"][:n], b["s"][:n])
            assert (st == want[:n]).all() and (bits == (want[:n] == 0)).all(), n
        # fused hash + verify over keys that have their 16-bit tables (3 000: two lanes per signature; 40 000: one)
        for n in (3000, 40000):
            rng = np.random.default_rng(18 + n)
            lens = rng.integers(0, 300, size=n)
            off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint32)
            arena = rng.integers(0, 256, size=int(off[-1]) + 1, dtype=np.uint8)
            fb = coracle.make_pool_batch(n, seed=9191, nkeys=nkeys, invalid_frac=0.2, digests=coracle.sha256_batch(arena, off))
            assert (fb["pool_qx"] == b["pool_qx"]).all()                              # (the pool depends on the seed only: the same six keys)
            bits, st = ctx.sha256_p256_verify_batch_keyed(arena, off, ids[fb["key_index"]], fb["r"], fb["s"])
            wantf = coracle.sha256_verify_batch(arena, off, fb["qx"], fb["qy"], fb["r"], fb["s"])
            assert (st == wantf).all() and (bits == (wantf == 0)).all(), n
        # the edge vectors (u1 = 0, final addition = doubling / infinity, window corners) through the 16-bit path
        vs = [v for v in _load("edge_kats.json") if len(v["e"]) == 64 and po.on_curve(int(v["qx"], 16), int(v["qy"], 16))
              and 0 <= int(v["r"], 16) < 1 << 256 and 0 <= int(v["s"], 16) < 1 << 256][:40]
        eids = np.array([ctx.key_register(_h32(v["qx"]), _h32(v["qy"])) for v in vs], dtype=np.uint32)
        assert ctx.test_key_tables16(int(eids[-1])) >= nkeys
        e = np.frombuffer(b"".join(_h32(v["e"]) for v in vs), dtype=np.uint8).reshape(-1, 32)
        r = np.frombuffer(b"".join(_h32(v["r"]) for v in vs), dtype=np.uint8).reshape(-1, 32)
        s_ = np.frombuffer(b"".join(_h32(v["s"]) for v in vs), dtype=np.uint8).reshape(-1, 32)
        qx = np.frombuffer(b"".join(_h32(v["qx"]) for v in vs), dtype=np.uint8).reshape(-1, 32)
        qy = np.frombuffer(b"".join(_h32(v["qy"]) for v in vs), dtype=np.uint8).reshape(-1, 32)
        bits, st = ctx.p256_verify_batch_keyed(eids, e, r, s_)
        assert (st == coracle.verify_batch(qx, qy, e, r, s_)).all()
        # more keys than 16-bit tables: keys beyond the 64th are served by their 8-bit combs, and so is every wavefront that meets one
        many = coracle.make_pool_batch(6000, seed=9393, nkeys=70, invalid_frac=0.1)
        mids = np.array([ctx.key_register(many["pool_qx"][j].tobytes(), many["pool_qy"][j].tobytes()) for j in range(70)], dtype=np.uint32)
        total = ctx.test_key_tables16(int(mids[0]))
        assert total == 64, total
        assert ctx.test_key_tables16(int(mids[-1])) == -1
        bits, st = ctx.p256_verify_batch_keyed(mids[many["key_index"]], many["e"], many["r"], many["s"])
        wantm = coracle.verify_batch(many["qx"], many["qy"], many["e"], many["r"], many["s"])
        assert (st == wantm).all() and (bits == (wantm == 0)).all()
    finally:
        ctx.close()


def test_key_tables_built_on_the_device_equal_the_host_builders_byte_for_byte(ctx):
    """Round 6: a registered key's comb table (T[w][d] = d 2^(8w) Q, 32 x 255 affine points in the fe29 Montgomery form of their canonical
    residues) is built by three kernels (csrc/keytab_kernels.hip) instead of 6 ms of host arithmetic.  Ground truth: the host builder
    (p256_tables29.h, unchanged - itself held against big-integer point arithmetic by tests/test_host_logic.py): all 163 840 words
    equal, for random keys, for the generator and 2G (a table the generator comb's tests know), and for keys registered in one batch
    by the provider.  Entry 0 of every window and the two pad words of every entry are zero; an independent spot check recomputes a
    few entries with the Python oracle."""
    b = coracle.make_pool_batch(8, seed=606, nkeys=5)
    keys = [(b["pool_qx"][j].tobytes(), b["pool_qy"][j].tobytes()) for j in range(5)]
    G = (po.GX.to_bytes(32, "big"), po.GY.to_bytes(32, "big"))
    g2 = po.pt_mul(2, (po.GX, po.GY))
    keys += [G, (g2[0].to_bytes(32, "big"), g2[1].to_bytes(32, "big"))]
    for qx, qy in keys:
        kid = ctx.key_register(qx, qy)
        dev = ctx.test_key_table(kid)
        host = fabgpu.Context.test_key_table_host(qx, qy)
        assert dev.shape == host.shape == (32 * 256 * 20,)
        assert np.array_equal(dev, host), "first differing word %d" % int(np.nonzero(dev != host)[0][0])
        t = dev.reshape(32, 256, 20)
        assert not t[:, 0, :].any() and not t[:, :, 18:].any()
    # the meaning of an entry, independently: limbs -> integer -> divide by R = 2^261 -> the affine coordinates of d 2^(8w) Q
    qx, qy = keys[0]
    t = ctx.test_key_table(ctx.key_register(qx, qy)).reshape(32, 256, 20)
    rinv = pow(1 << 261, -1, po.P)

    def plain(limbs):
        return sum(int(v) << (29 * i) for i, v in enumerate(limbs)) * rinv % po.P
    Q = (int.from_bytes(qx, "big"), int.from_bytes(qy, "big"))
    for w, d in ((0, 1), (0, 255), (7, 128), (31, 3), (31, 255), (16, 77)):
        want = po.pt_mul(d << (8 * w), Q)
        assert (plain(t[w, d, :9]), plain(t[w, d, 9:18])) == want, (w, d)


def test_generator_comb_built_on_the_device_equals_the_host_builders(ctx):
    """fabgpu_init builds the generator's 16-bit comb (16 windows x 65 535 affine points, 80 MiB) on the device since round 6
    (csrc/keytab_kernels.hip launch_gtab_build) instead of on sixteen host threads: all 20 971 520 words equal the host builder's
    (p256_tables29.h, the table every verify kernel gathered from through round 5).  And a context made with the host builder
    (FABGPU_GTAB_HOST=1, read once at fabgpu_init) agrees with itself the same way."""
    import time
    assert ctx.test_gtab_compare_with_host() == -1
    os.environ["FABGPU_GTAB_HOST"] = "1"
    try:
        t0 = time.perf_counter()
        c2 = fabgpu.Context(device=0)
        host_s = time.perf_counter() - t0
    finally:
        del os.environ["FABGPU_GTAB_HOST"]
    try:
        assert c2.test_gtab_compare_with_host() == -1
    finally:
        c2.close()
    t0 = time.perf_counter()
    c3 = fabgpu.Context(device=0)
    dev_s = time.perf_counter() - t0
    c3.close()
    print("fabgpu_init: %.0f ms with the generator comb built on the device, %.0f ms with the host builder" % (dev_s * 1e3, host_s * 1e3))
    assert dev_s < host_s


def test_registered_keys_every_signer_its_own_key_edge_vectors_and_bad_ids(ctx):
    # the edge vectors (x(R) >= n, u1 = 0, final addition = doubling / infinity, window corners ...) through the keyed path
    vs = [v for v in _load("edge_kats.json") if len(v["e"]) == 64 and po.on_curve(int(v["qx"], 16), int(v["qy"], 16))
          and 0 <= int(v["r"], 16) < 1 << 256 and 0 <= int(v["s"], 16) < 1 << 256]
    assert len(vs) > 60
    ids = np.array([ctx.key_register(_h32(v["qx"]), _h32(v["qy"])) for v in vs], dtype=np.uint32)
    bits, st = ctx.p256_verify_batch_keyed(ids, *[_arr([_h32(v[k]) for v in vs]) for k in ("e", "r", "s")])
    for v, b, s in zip(vs, bits, st):
        assert s == v["status"] and b == (v["status"] == 0), v["name"]
    # off-curve keys are refused at registration (the KeyImport gate), unknown ids come back as status 4, never as a verdict
    off = [v for v in _load("edge_kats.json") if not po.on_curve(int(v["qx"], 16), int(v["qy"], 16)) and int(v["qx"], 16) < 1 << 256 and int(v["qy"], 16) < 1 << 256]
    assert off
    for v in off:
        with pytest.raises(fabgpu.FabgpuError):
            ctx.key_register(_h32(v["qx"]), _h32(v["qy"]))
    b = coracle.make_batch(5, seed=9)
    kid = ctx.key_register(b["qx"][0].tobytes(), b["qy"][0].tobytes())
    bits, st = ctx.p256_verify_batch_keyed(np.array([kid, 0xFFFFFFF0, kid + 100000], dtype=np.uint32), b["e"][:3], b["r"][:3], b["s"][:3])
    assert list(st) == [0, 4, 4] and list(bits) == [True, False, False]


_EMPTY_LANE_ROWS = ([], [])


def test_registered_keys_scalars_whose_digits_leave_lanes_empty(ctx):
    """The eight-lanes-per-signature kernels (p256_wide29.h) add a comb's table entries up lane by lane - u2's 32 eight-bit windows
    four to a lane, u1's 16 sixteen-bit windows two to a lane - and then across lanes.  VALID signatures whose scalars make whole
    lanes (or all but one, or every lane) contribute NOTHING - partial sums at infinity at every level of the tree - and scalars at
    the top of the range; every one with a registered key of its own, each beside a broken twin.  A signer can hit any (u1, u2): for a
    nonce k, r = x(kG), s = r / u2, e = u1 s, d = (s k - e) / r.  (Run by every context configuration: the two-lane and one-lane keyed
    kernels must say the same.)"""
    rng = np.random.default_rng(44)
    N = po.N

    def lane_mask(lanes, bits, per):                                          # all-ones digits in the windows of the given lanes
        v = 0
        for ln in lanes:
            for w in range(per * ln, per * ln + per):
                v |= ((1 << bits) - 1) << (bits * w)
        return v % N or 1
    u2s = [1, 0xFF, 1 << 8, 1 << 31, 1 << 32, 1 << 248, (1 << 248) | 1, lane_mask([0], 8, 4), lane_mask([7], 8, 4), lane_mask([0, 7], 8, 4),
           lane_mask([1, 2], 8, 4), lane_mask([3], 8, 4), lane_mask([0, 2, 4, 6], 8, 4), lane_mask([1, 3, 5, 7], 8, 4), N - 1, N - 2, N >> 1, (N >> 1) + 1,
           int("01" * 32, 16), int("0100" * 16, 16), int("00000001" * 8, 16), int("ff000000" * 8, 16) % N]
    u1s = [0, 1, 0xFFFF, 1 << 16, 1 << 240, (1 << 240) | 1, lane_mask([0], 16, 2), lane_mask([7], 16, 2), lane_mask([3, 4], 16, 2),
           lane_mask([0, 2, 4, 6], 16, 2), N - 1, N - 2, int("0001" * 16, 16), int("00010000" * 8, 16), int("ffff0000" * 8, 16) % N]
    rows, want = _EMPTY_LANE_ROWS
    for u2 in ([] if rows else u2s):                                         # (built once: every context configuration runs the same rows)
        for u1 in u1s + [int(rng.integers(1, 1 << 62)) ** 4 % N]:
            for _ in range(400):
                k = int(rng.integers(1, 1 << 62)) * int(rng.integers(1, 1 << 62)) + 1
                R = po.pt_mul(k, (po.GX, po.GY))
                r = R[0] % N
                s = r * pow(u2, -1, N) % N
                if r == 0 or s == 0 or not po.is_low_s(s):
                    continue
                e = u1 * s % N
                d = (s * k - e) * pow(r, -1, N) % N
                if d == 0:
                    continue
                Q = po.pt_mul(d, (po.GX, po.GY))
                digest = e.to_bytes(32, "big")
                assert po.ecdsa_verify_raw(Q[0], Q[1], digest, r, s), (hex(u1), hex(u2))
                rows.append((Q, digest, r, s))
                want.append(0)
                rows.append((Q, digest, r, (s + 1) % N if po.is_low_s((s + 1) % N) and (s + 1) % N else s - 1))
                want.append(None)
                break
            else:
                raise AssertionError("no signature for u1 = %x, u2 = %x" % (u1, u2))
    ids = np.array([ctx.key_register(_h32("%064x" % Q[0]), _h32("%064x" % Q[1])) for Q, _, _, _ in rows], dtype=np.uint32)
    cols = [_arr([dg for _, dg, _, _ in rows]), _arr([_h32("%064x" % r) for _, _, r, _ in rows]), _arr([_h32("%064x" % s) for _, _, _, s in rows])]
    oracle = coracle.verify_batch(_arr([_h32("%064x" % Q[0]) for Q, _, _, _ in rows]), _arr([_h32("%064x" % Q[1]) for Q, _, _, _ in rows]), *cols)
    for w, o in zip(want, oracle):
        assert w is None or o == w
    for lo in (0, 5):                                                        # (a batch that starts inside a wavefront's eight signatures, too)
        bits, st = ctx.p256_verify_batch_keyed(ids[lo:], *[c[lo:] for c in cols])
        assert (st == oracle[lo:]).all() and (bits == (oracle[lo:] == 0)).all(), np.nonzero(st != oracle[lo:])[0][:8]
    assert int((oracle == 0).sum()) >= len(u2s) * (len(u1s) + 1) and int((oracle != 0).sum()) >= len(u2s) * len(u1s)


def test_registered_keys_fused_hash_verify_vs_oracle(ctx):
    n = 3000
    rng = np.random.default_rng(18)
    lens = rng.integers(0, 2500, size=n)
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint32)
    arena = rng.integers(0, 256, size=int(off[-1]) + 1, dtype=np.uint8)
    b = coracle.make_pool_batch(n, seed=19, nkeys=8, invalid_frac=0.2, digests=coracle.sha256_batch(arena, off))
    ids = np.array([ctx.key_register(b["pool_qx"][j].tobytes(), b["pool_qy"][j].tobytes()) for j in range(8)], dtype=np.uint32)
    bits, st = ctx.sha256_p256_verify_batch_keyed(arena, off, ids[b["key_index"]], b["r"], b["s"])
    want = coracle.sha256_verify_batch(arena, off, b["qx"], b["qy"], b["r"], b["s"])
    assert (st == want).all() and (bits == (want == 0)).all()
    bits2, st2 = ctx.sha256_p256_verify_batch(arena, off, b["qx"], b["qy"], b["r"], b["s"])
    assert (st2 == st).all() and (bits2 == bits).all()


@pytest.mark.parametrize("n", [1500, 2600])        # (keyed: eight lanes per message and whole-message hashing up to 2 048; mid-states beyond)
@pytest.mark.parametrize("keyed", [False, True])
def test_identity_batch_with_shared_prefixes_vs_hashlib_and_oracle(ctx, keyed, n):
    """fabgpu_identity_verify_batch: message = prefix || suffix with the prefix hashed once (mid-state).  Prefix lengths
    around every multiple of 64 (so the re-read tail is 0..63 bytes), arbitrary alignments, empty suffixes, messages without
    a prefix, prefixes nobody uses.  The digests are pinned by hashlib on the concatenation, the verdicts by the oracle and by
    the un-prefixed fused kernel on the materialised messages."""
    rng = np.random.default_rng(64)
    plens = [0, 1, 55, 56, 63, 64, 65, 119, 120, 127, 128, 129, 191, 192, 1023, 1024, 1025, 1856] + [int(x) for x in rng.integers(0, 300, size=12)]
    m = len(plens)
    pre_idx = rng.integers(0, m, size=n).astype(np.uint32)
    pre_idx[rng.random(n) < 0.1] = 0xFFFFFFFF
    slens = rng.integers(0, 200, size=n)
    slens[rng.random(n) < 0.05] = 0
    # arena: a little junk, the prefixes back to back, more junk, the suffixes
    parts, pre_off, off = [bytes(rng.integers(0, 256, size=3, dtype=np.uint8))], [], []
    pos = 3
    prefixes = []
    for L in plens:
        p = bytes(rng.integers(0, 256, size=L, dtype=np.uint8)); prefixes.append(p)
        pre_off.append(pos); parts.append(p); pos += L
    pre_off.append(pos)
    parts.append(b"\xAA" * 5); pos += 5
    suffixes = []
    for L in slens:
        sfx = bytes(rng.integers(0, 256, size=int(L), dtype=np.uint8)); suffixes.append(sfx)
        off.append(pos); parts.append(sfx); pos += int(L)
    off.append(pos)
    arena = np.frombuffer(b"".join(parts) + b"\0", dtype=np.uint8)
    msgs = [(prefixes[pi] if pi != 0xFFFFFFFF else b"") + sfx for pi, sfx in zip(pre_idx.tolist(), suffixes)]
    dig = np.frombuffer(b"".join(hashlib.sha256(mm).digest() for mm in msgs), dtype=np.uint8).reshape(n, 32)
    b = coracle.make_pool_batch(n, seed=65, nkeys=6, invalid_frac=0.2, digests=dig)
    want = coracle.verify_batch(b["qx"], b["qy"], b["e"], b["r"], b["s"])          # e = the hashlib digest (or its mutation)
    want_hash = want.copy()
    want_hash[b["kind"] == 1] = 0                                                   # kind 1 flipped e only: the message decides
    kw = {}
    if keyed:
        ids = np.array([ctx.key_register(b["pool_qx"][j].tobytes(), b["pool_qy"][j].tobytes()) for j in range(6)], dtype=np.uint32)
        kw["key_id"] = ids[b["key_index"]]
    else:
        kw.update(qx=b["qx"], qy=b["qy"])
    bits, st = ctx.identity_verify_batch(arena, np.array(off, dtype=np.uint32), b["r"], b["s"], pre_off=np.array(pre_off, dtype=np.uint32),
                                         pre_idx=pre_idx, **kw)
    assert (st == want_hash).all() and (bits == (want_hash == 0)).all()
    # the same messages materialised, through the plain fused entry point
    moff = np.concatenate([[0], np.cumsum([len(x) for x in msgs])]).astype(np.uint32)
    marena = np.frombuffer(b"".join(msgs) + b"\0", dtype=np.uint8)
    bits2, st2 = ctx.sha256_p256_verify_batch(marena, moff, b["qx"], b["qy"], b["r"], b["s"])
    assert (st2 == st).all() and (bits2 == bits).all()
    # and without any prefix the described entry point equals the plain one
    bits3, st3 = ctx.identity_verify_batch(marena, moff, b["r"], b["s"], **kw)
    assert (st3 == st).all()
    # span mode (FABGPU_IDB_SPANS): the same messages as (start, end) pairs in a permuted order, prefixes as pairs too
    perm = rng.permutation(n)
    offa = np.array(off, dtype=np.uint32)
    spans = np.stack([offa[:-1][perm], offa[1:][perm]], axis=1).reshape(-1)
    poa = np.array(pre_off, dtype=np.uint32)
    pspans = np.stack([poa[:-1], poa[1:]], axis=1).reshape(-1)
    kw2 = {k: v[perm] for k, v in kw.items()}
    bits4, st4 = ctx.identity_verify_batch(arena, spans, b["r"][perm], b["s"][perm], pre_off=pspans, pre_idx=pre_idx[perm], spans=True, **kw2)
    assert (st4 == st[perm]).all() and (bits4 == bits[perm]).all()
    # gathered hashes riding along (the TxID / proposal-hash checks of the block pass): messages stitched from up to three spans
    g = []
    want_g = []
    for j in range(50):
        pieces = []
        for _ in range(int(rng.integers(0, 4))):
            a0 = int(rng.integers(0, arena.size - 1))
            pieces.append((a0, min(arena.size - 1, a0 + int(rng.integers(0, 400)))))
        while len(pieces) < 3:
            pieces.insert(int(rng.integers(0, len(pieces) + 1)), (7, 7))                  # unused piece: start == end
        g.append([x for pr in pieces for x in pr])
        want_g.append(hashlib.sha256(b"".join(arena[a0:a1].tobytes() for a0, a1 in pieces)).digest())
    bits5, st5, dig5 = ctx.identity_verify_batch(arena, spans, b["r"][perm], b["s"][perm], pre_off=pspans, pre_idx=pre_idx[perm], spans=True,
                                                 gather_spans=np.array(g, dtype=np.uint32), **kw2)
    assert (st5 == st4).all() and [d.tobytes() for d in dig5] == want_g
    # the arena staged ahead (fabgpu_arena_stage): same answers; a token that a newer upload replaced is refused
    tok = ctx.arena_stage(arena)
    bits6, st6, dig6 = ctx.identity_verify_batch(arena, spans, b["r"][perm], b["s"][perm], pre_off=pspans, pre_idx=pre_idx[perm], spans=True,
                                                 gather_spans=np.array(g, dtype=np.uint32), stage_token=tok, **kw2)
    assert (st6 == st4).all() and (bits6 == bits4).all() and [d.tobytes() for d in dig6] == want_g
    # the context keeps its three most recent uploads (channels validating at once): one newer upload leaves the token good ...
    tok2 = ctx.arena_stage(arena[:100])
    assert tok2 != tok
    bits7, st7, _ = ctx.identity_verify_batch(arena, spans, b["r"][perm], b["s"][perm], pre_off=pspans, pre_idx=pre_idx[perm], spans=True,
                                              gather_spans=np.array(g, dtype=np.uint32), stage_token=tok, **kw2)
    assert (st7 == st4).all() and (bits7 == bits4).all()
    # ... and the short upload is refused for spans beyond its length; two more uploads push the first one out
    with pytest.raises(fabgpu.FabgpuError):
        ctx.identity_verify_batch(arena, spans, b["r"][perm], b["s"][perm], pre_off=pspans, pre_idx=pre_idx[perm], spans=True, stage_token=tok2, **kw2)
    toks = {tok, tok2, ctx.arena_stage(arena[:200]), ctx.arena_stage(arena[:300])}
    assert len(toks) == 4
    with pytest.raises(fabgpu.FabgpuError):
        ctx.identity_verify_batch(arena, spans, b["r"][perm], b["s"][perm], pre_off=pspans, pre_idx=pre_idx[perm], spans=True, stage_token=tok, **kw2)


# ---- SHA-256 ----------------------------------------------------------------------------------------
def test_sha256_like_reference_TestSHA(ctx):
    rng = np.random.default_rng(5)
    lens = list(range(0, 130)) + [183, 184, 191, 192, 1023, 1024, 1855, 1856, 1857, 4608, 5000]
    msgs = [rng.integers(0, 256, size=l, dtype=np.uint8).tobytes() for l in lens]
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint32)
    arena = np.frombuffer(b"".join(msgs) + b"\0" * 3, dtype=np.uint8)
    d = ctx.sha256_batch(arena, off)
    for i, m in enumerate(msgs):
        assert d[i].tobytes() == hashlib.sha256(m).digest(), lens[i]
    assert (d == coracle.sha256_batch(arena, off)).all()


def test_sha256_both_roads_agree_on_every_length_and_alignment(ctx):
    """Up to 2 048 messages a launch puts eight lanes on a message (sha256_coop.h: lengths that end a block, spill the length field into
    a block of its own, span several eight-block chunks, start at every byte alignment); beyond, one lane per message.  The same
    messages both ways, against hashlib."""
    rng = np.random.default_rng(55)
    lens = [0, 1, 3, 4, 54, 55, 56, 57, 63, 64, 65, 119, 120, 121, 127, 128, 447, 448, 449, 503, 504, 511, 512, 513, 1023, 1024, 1087, 3391, 3392, 3400, 4608]
    parts, off, pos = [], [], 0
    for k, L in enumerate(lens * 4):
        pad = k % 4                                       # every alignment for every length
        parts.append(b"\x55" * pad); pos += pad
        off.append(pos)
        parts.append(rng.integers(0, 256, size=L, dtype=np.uint8).tobytes()); pos += L
        off.append(pos)
    arena = np.frombuffer(b"".join(parts) + b"\0" * 8, dtype=np.uint8)
    spans = np.array(off, dtype=np.uint32).reshape(-1, 2)
    want = [hashlib.sha256(arena[a:b].tobytes()).digest() for a, b in spans]
    # n + 1 consecutive offsets are what the flat entry point takes: hash the gaps too and ignore them
    flat = np.array(sorted(set(off)), dtype=np.uint32)
    idx = {int(a): i for i, a in enumerate(flat[:-1])}
    small = ctx.sha256_batch(arena, flat)
    assert len(flat) - 1 <= 2048
    for (a, b), w in zip(spans, want):
        if b > a:
            assert small[idx[int(a)]].tobytes() == w, (a, b)
    # the same list in disjoint copies of the arena (each copy's last message - the arena's tail and the next copy's padding - is a dummy)
    per = len(flat)
    reps = 2048 // per + 1
    big_arena = np.tile(arena, reps)
    big_off = np.concatenate([flat.astype(np.uint64) + r * len(arena) for r in range(reps)] + [np.array([reps * len(arena)], dtype=np.uint64)]).astype(np.uint32)
    assert len(big_off) - 1 > 2048
    big = ctx.sha256_batch(big_arena, big_off)
    for r in (0, reps - 1):
        for (a, b), w in zip(spans, want):
            if b > a:
                assert big[r * per + idx[int(a)]].tobytes() == w, (r, a, b)


def test_sha256_misaligned_overlapping_and_offset_base(ctx):
    rng = np.random.default_rng(6)
    arena = rng.integers(0, 256, size=5000, dtype=np.uint8)
    for base in (0, 1, 2, 3, 77):
        off = np.array([base, base + 5, base + 5, base + 70, base + 1000, base + 1000 + 1856], dtype=np.uint32)
        d = ctx.sha256_batch(arena, off)
        for i in range(5):
            assert d[i].tobytes() == hashlib.sha256(arena[off[i]:off[i + 1]].tobytes()).digest()


def test_sha256_block_of_identical_lengths(ctx):
    n, L = 3000, 1856                                   # prp(1024) || endorser(832), SURVEY 8(d)
    arena = np.random.default_rng(7).integers(0, 256, size=n * L, dtype=np.uint8)
    off = (np.arange(n + 1, dtype=np.uint64) * L).astype(np.uint32)
    assert (ctx.sha256_batch(arena, off) == coracle.sha256_batch(arena, off)).all()


# ---- fused identity.Verify ---------------------------------------------------------------------------
def test_fused_hash_verify_vs_oracle(ctx):
    n = 2000
    rng = np.random.default_rng(8)
    lens = rng.integers(0, 2500, size=n)
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint32)
    arena = rng.integers(0, 256, size=int(off[-1]) + 1, dtype=np.uint8)
    dig = coracle.sha256_batch(arena, off)
    b = coracle.make_batch(n, seed=9, invalid_frac=0.2, digests=dig)
    bits, st = ctx.sha256_p256_verify_batch(arena, off, b["qx"], b["qy"], b["r"], b["s"])
    want = coracle.sha256_verify_batch(arena, off, b["qx"], b["qy"], b["r"], b["s"])
    assert (st == want).all() and (bits == (want == 0)).all()
    assert (want == coracle.sha256_verify_batch(arena, off, b["qx"], b["qy"], b["r"], b["s"], use_ossl=True)).all()
    # round trip through the two-kernel path: digests from the GPU, then verify-only
    bits2, st2 = ctx.p256_verify_batch(b["qx"], b["qy"], ctx.sha256_batch(arena, off), b["r"], b["s"])
    assert (st2 == st).all()


# ---- the reference's own test shapes, through the BCCSP mirror -----------------------------------------
def _keypair(seed):
    d = 1 + seed * 7919
    return d, fabgpu.ECDSAPublicKey(*po.pt_mul(d, (po.GX, po.GY)))


def test_TestECDSAVerify_and_msp_sign_verify_tamper(csp):
    d, pk = _keypair(1)
    msg = b"Hello World"
    digest = csp.hash(msg, fabgpu.SHA256Opts())
    assert digest == hashlib.sha256(msg).digest()
    sig = po.marshal_ecdsa_signature(*po.sign_raw(d, digest, 0xABCDEF123))
    assert csp.verify(pk, sig, digest) is True
    assert csp.verify(pk, sig, csp.hash(msg + b"!", fabgpu.SHA256Opts())) is False
    idn = fabgpu.Identity(csp, pk)
    assert idn.verify(msg, sig) is None
    with pytest.raises(fabgpu.BCCSPError, match="The signature is invalid"):
        idn.verify(msg + b"x", sig)                                   # msp/msp_test.go:532-535
    other = fabgpu.Identity(csp, _keypair(2)[1])
    with pytest.raises(fabgpu.BCCSPError, match="The signature is invalid"):
        other.verify(msg, sig)


def test_TestECDSALowS(csp):
    d, pk = _keypair(3)
    digest = hashlib.sha256(b"Hello World").digest()
    r, s = po.sign_raw(d, digest, 0x1234567, low_s=True)
    assert s <= po.HALF_N and csp.verify(pk, po.marshal_ecdsa_signature(r, s), digest)
    with pytest.raises(fabgpu.BCCSPError, match=r"Invalid S. Must be smaller than half the order \[%d\]\[%d\]" % (po.N - s, po.HALF_N)):
        csp.verify(pk, po.marshal_ecdsa_signature(r, po.N - s), digest)   # bccsp/sw/ecdsa_test.go:66-73
    by = {v["name"]: v for v in _load("edge_kats.json")}
    for name, ok in (("s_eq_half_n", True), ("s_eq_half_n_plus_1", False)):
        v = by[name]
        k = fabgpu.ECDSAPublicKey(int(v["qx"], 16), int(v["qy"], 16))
        sig = po.marshal_ecdsa_signature(int(v["r"], 16), int(v["s"], 16))
        if ok:
            assert csp.verify(k, sig, bytes.fromhex(v["e"])) is True
        else:
            with pytest.raises(fabgpu.BCCSPError, match="Invalid S"):
                csp.verify(k, sig, bytes.fromhex(v["e"]))


def test_TestECDSASignatureEncoding_and_argument_errors(csp):
    _, pk = _keypair(4)
    for v in _load("der_kats.json"):
        if v["ok"]:
            continue
        raw = bytes.fromhex(v["der"])
        if not raw:
            continue
        with pytest.raises(fabgpu.BCCSPError, match=r"Failed verifing with opts \[<nil>\]: Failed unmashalling signature \["):
            csp.verify(pk, raw, b"\x01" * 32)
    with pytest.raises(fabgpu.BCCSPError, match="Invalid Key. It must not be nil."):
        csp.verify(None, b"\x01", b"\x01")                               # bccsp/sw/impl.go:249-257
    with pytest.raises(fabgpu.BCCSPError, match="Invalid signature. Cannot be empty."):
        csp.verify(pk, b"", b"\x01")
    with pytest.raises(fabgpu.BCCSPError, match="Invalid digest. Cannot be empty."):
        csp.verify(pk, b"\x30\x06\x02\x01\x01\x02\x01\x01", b"")
    with pytest.raises(fabgpu.BCCSPError, match="Invalid opts. It must not be nil."):
        csp.hash(b"x", None)                                             # bccsp/sw/impl.go:179-181
    with pytest.raises(fabgpu.BCCSPError, match="Unsupported 'HashOpt' provided"):
        csp.hash(b"x", fabgpu.SHA3_256Opts())


def test_csp_verify_equals_oracle_csp_verify_on_der_and_digest_shapes(csp):
    # every DER vector x several digest lengths: (valid, error-class) identical to the restated bccsp/sw
    _, pk0 = _keypair(5)
    keys, sigs, digs, want = [], [], [], []
    for v in _load("der_kats.json"):
        raw = bytes.fromhex(v["der"])
        for dg in (b"\x01" * 32, b"\x07", b"\xff" * 40):
            keys.append(pk0); sigs.append(raw); digs.append(dg)
    for v in _load("edge_kats.json"):
        r, s = int(v["r"], 16), int(v["s"], 16)
        if po.on_curve(int(v["qx"], 16), int(v["qy"], 16)):
            keys.append(fabgpu.ECDSAPublicKey(int(v["qx"], 16), int(v["qy"], 16)))
            sigs.append(po.marshal_ecdsa_signature(r, s)); digs.append(bytes.fromhex(v["e"]))
    for k, sg, dg in zip(keys, sigs, digs):
        try:
            want.append((po.csp_verify((k.x, k.y), sg, dg), None))
        except po.BCCSPError as e:
            want.append((False, str(e)))
    got = csp.verify_batch(keys, sigs, digs)
    for (gv, ge), (wv, we), sg in zip(got, want, sigs):
        assert gv == wv, sg.hex()
        assert (ge is None) == (we is None), (ge, we)
        if we is not None:
            for needle in ("Invalid S. Must be smaller", "Failed unmashalling signature [", "Cannot be empty", "larger than zero"):
                assert (needle in ge) == (needle in we), (ge, we)
            if "Invalid S" in we:
                assert ge == we


def test_coalesced_one_signature_calls_from_many_threads(csp):
    """bccsp.Verify / identity.Verify one signature at a time from 48 threads (orderer Broadcast handlers, validator goroutines):
    fabgpu_csp_verify_coalesced / fabgpu_csp_identity_verify_coalesced give every caller the answer - and the error text - of the
    batch entry points, which the tests above hold against the restated bccsp/sw; calls in flight together share launches."""
    import threading
    _, pk0 = _keypair(5)
    keys, sigs, digs = [], [], []
    for v in _load("der_kats.json"):
        keys.append(pk0); sigs.append(bytes.fromhex(v["der"])); digs.append(b"\x01" * 32)
    for v in _load("edge_kats.json"):
        if po.on_curve(int(v["qx"], 16), int(v["qy"], 16)):
            keys.append(fabgpu.ECDSAPublicKey(int(v["qx"], 16), int(v["qy"], 16)))
            sigs.append(po.marshal_ecdsa_signature(int(v["r"], 16), int(v["s"], 16))); digs.append(bytes.fromhex(v["e"]))
    rng = np.random.default_rng(21)
    for t in range(400):                                   # fresh signatures, a third of them tampered
        d, pk = _keypair(100 + t % 7)
        dg = bytes(rng.integers(0, 256, size=32, dtype=np.uint8))
        r, s_ = po.sign_raw(d, dg, int(rng.integers(1, 1 << 62)))
        if t % 3 == 0:
            dg = bytes([dg[0] ^ 1]) + dg[1:]
        keys.append(pk); sigs.append(po.marshal_ecdsa_signature(r, s_)); digs.append(dg)
    want = csp.verify_batch(keys, sigs, digs)
    n = len(keys)
    before = csp.coalescer_stats()
    got = [None] * n
    errors = []

    def worker(w, nw):
        try:
            for i in range(w, n, nw):
                try:
                    got[i] = (csp.verify_coalesced(keys[i], sigs[i], digs[i]), None)
                except fabgpu.BCCSPError as e:
                    got[i] = (False, str(e))
        except Exception:                                  # noqa: BLE001
            import traceback
            errors.append(traceback.format_exc())
    th = [threading.Thread(target=worker, args=(w, 48)) for w in range(48)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=300)
    assert not errors, errors[0]
    assert got == [(v, e) for v, e in want]
    after = csp.coalescer_stats()
    calls, launches = after["calls"] - before["calls"], after["launches"] - before["launches"]
    assert 0 < launches < calls / 3 and after["largest_batch"] >= 4, (calls, launches, after)
    # identity.Verify (hash fused on the device), messages of ragged lengths
    d, pk = _keypair(6)
    msgs = [bytes(rng.integers(0, 256, size=int(rng.integers(0, 3000)), dtype=np.uint8)) for _ in range(300)]
    isigs = [po.marshal_ecdsa_signature(*po.sign_raw(d, hashlib.sha256(m).digest(), 1000 + j)) for j, m in enumerate(msgs)]
    for j in range(0, 300, 4):
        msgs[j] = msgs[j] + b"!"
    iwant = csp.identity_verify_batch([pk] * 300, msgs, isigs)
    igot = [None] * 300

    def iworker(w, nw):
        try:
            for i in range(w, 300, nw):
                igot[i] = csp.identity_verify_coalesced(pk, msgs[i], isigs[i])
        except Exception:                                  # noqa: BLE001
            import traceback
            errors.append(traceback.format_exc())
    th = [threading.Thread(target=iworker, args=(w, 32)) for w in range(32)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=300)
    assert not errors, errors[0]
    assert igot == list(iwant) and sum(1 for e in igot if e is None) == 225
    assert csp.identity_verify_coalesced(None, b"m", isigs[0]) == "could not determine the validity of the signature: Invalid Key. It must not be nil."


def test_fullflow_every_single_byte_mutation_flips_the_verdict(csp):
    # core/common/validation/fullflow_test.go:240-250: for i := range payload { payload[i]++ ... must fail }
    d, pk = _keypair(6)
    payload = bytes(np.random.default_rng(10).integers(0, 256, size=700, dtype=np.uint8))
    sig = po.marshal_ecdsa_signature(*po.sign_raw(d, hashlib.sha256(payload).digest(), 0x77777))
    msgs = [payload] + [payload[:i] + bytes([(payload[i] + 1) & 0xFF]) + payload[i + 1:] for i in range(len(payload))]
    errs = csp.identity_verify_batch([pk] * len(msgs), msgs, [sig] * len(msgs))
    assert errs[0] is None
    assert all(e == "The signature is invalid" for e in errs[1:])
    # ... and every single-byte mutation of the signature fails one way or the other
    sigs = [sig[:i] + bytes([(sig[i] + 1) & 0xFF]) + sig[i + 1:] for i in range(len(sig))]
    errs = csp.identity_verify_batch([pk] * len(sigs), [payload] * len(sigs), sigs)
    want = [po.identity_verify((pk.x, pk.y), payload, s) for s in sigs]
    for e, w in zip(errs, want):
        assert (e is None) == (w is None)
        assert e is None or e.split(":")[0] == w.split(":")[0]


def test_block_endorsement_prepass(csp):
    # validator_keylevel.go:246-258 message construction + verify-all-then-evaluate
    rng = np.random.default_rng(11)
    endorsers = []
    for j in range(4):
        d, pk = _keypair(20 + j)
        endorsers.append((d, pk, bytes(rng.integers(0, 256, size=832, dtype=np.uint8))))
    txs, want = [], []
    for t in range(40):
        prp = bytes(rng.integers(0, 256, size=1024, dtype=np.uint8))
        ends, ok = [], True
        for j in rng.choice(4, size=3, replace=False):
            d, pk, ident = endorsers[j]
            r, s = po.sign_raw(d, hashlib.sha256(prp + ident).digest(), int(rng.integers(1, 1 << 62)))
            if t % 7 == 3 and j == 1:
                r += 1; ok = False
            ends.append((ident, pk, po.marshal_ecdsa_signature(r, s)))
        txs.append((prp, ends)); want.append(ok)
    assert (fabgpu.validate_block_endorsements(csp, txs) == np.array(want)).all()
    # the same block once the endorsers came in through BCCSP.KeyImport: every signer is registered, the pass runs on the
    # keyed fused kernel (fabgpu_sha256_p256_verify_batch_keyed) and must say exactly the same
    before = csp.key_count()
    for d, pk, ident in endorsers:
        csp.key_import((pk.x, pk.y))
    assert csp.key_count() == before + 4
    assert (fabgpu.validate_block_endorsements(csp, txs) == np.array(want)).all()
    # ... and a block with one endorser that was never imported falls back to the fresh-key kernel, same verdicts
    d5, pk5 = _keypair(77)
    ident5 = bytes(rng.integers(0, 256, size=832, dtype=np.uint8))
    prp = bytes(rng.integers(0, 256, size=1024, dtype=np.uint8))
    r, s = po.sign_raw(d5, hashlib.sha256(prp + ident5).digest(), 12345)
    txs2 = txs + [(prp, [(ident5, pk5, po.marshal_ecdsa_signature(r, s))])]
    assert (fabgpu.validate_block_endorsements(csp, txs2) == np.array(want + [True])).all()


def test_device_resident_entry_points_on_torch_stream(ctx):
    import torch
    n = 5000
    b = fabgpu.synth_batch(n, seed=12, invalid_permille=50)
    t = {k: torch.from_numpy(b[k]).cuda() for k in ("qx", "qy", "e", "r", "s")}
    words = torch.zeros((n + 63) // 64, dtype=torch.int64, device="cuda")
    status = torch.full((n,), 9, dtype=torch.uint8, device="cuda")
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        ctx.p256_verify_batch_dev(n, t["qx"].data_ptr(), t["qy"].data_ptr(), t["e"].data_ptr(), t["r"].data_ptr(), t["s"].data_ptr(),
                                  words.data_ptr(), status.data_ptr(), s.cuda_stream)
    s.synchronize()
    want = coracle.verify_batch(b["qx"], b["qy"], b["e"], b["r"], b["s"])
    assert (status.cpu().numpy() == want).all()
    assert (fabgpu.unpack_bits(words.cpu().numpy().view(np.uint64), n) == (want == 0)).all()
    assert ctx.last_kernel_ms() > 0


def test_concurrent_launches_on_two_streams_get_separate_workspaces(ctx):
    """bccsp.Verify is called from many goroutines (v20/validator.go:198-208): two host threads launch on two streams at
    once; each launch must borrow its own per-lane-table workspace from the pool."""
    import threading

    import torch
    results = {}

    def worker(tag, seed, n):
        torch.cuda.set_device(0)
        st = torch.cuda.Stream()
        b = fabgpu.synth_batch(n, seed=seed, invalid_permille=100)
        dev = {k: torch.from_numpy(b[k]).cuda() for k in ("qx", "qy", "e", "r", "s")}
        words = torch.zeros((n + 63) // 64, dtype=torch.int64, device="cuda")
        torch.cuda.synchronize()
        ok = True
        for _ in range(6):
            words.zero_()
            st.wait_stream(torch.cuda.current_stream())
            ctx.p256_verify_batch_dev(n, dev["qx"].data_ptr(), dev["qy"].data_ptr(), dev["e"].data_ptr(), dev["r"].data_ptr(),
                                      dev["s"].data_ptr(), words.data_ptr(), 0, st.cuda_stream)
            st.synchronize()
            got = fabgpu.unpack_bits(words.cpu().numpy().view(np.uint64), n)
            ok = ok and bool((got == (b["kind"] == 0)).all())
        results[tag] = ok
    ths = [threading.Thread(target=worker, args=("a", 101, 20000)), threading.Thread(target=worker, args=("b", 202, 70000))]
    for th in ths:
        th.start()
    for th in ths:
        th.join()
    assert results == {"a": True, "b": True}


# ---- the failure contract: a device fault is an infrastructure error, never a verdict (SURVEY section 5 "determinism under failure") ----
_FAULT_SNIPPET = r"""
import ctypes, sys, numpy as np
sys.path[:0] = [%(root)r + "/fabric-mod_amd", %(root)r + "/oracle"]
import fabgpu
L = fabgpu.load()
ctx = fabgpu.Context(device=0)
n = 300
b = fabgpu.synth_batch(n, seed=3, invalid_permille=100)
u8 = ctypes.POINTER(ctypes.c_uint8)
p = lambda a: a.ctypes.data_as(u8)
bits = np.full((n + 63) // 64, 0xA5A5A5A5A5A5A5A5, dtype=np.uint64)          # sentinels: a failed call must not touch them
st = np.full(n, 0xEE, dtype=np.uint8)
rc = L.fabgpu_p256_verify_batch(ctx.handle, n, p(b["qx"]), p(b["qy"]), p(b["e"]), p(b["r"]), p(b["s"]), bits.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)), p(st))
assert rc == %(want)d, rc
assert (bits == 0xA5A5A5A5A5A5A5A5).all() and (st == 0xEE).all(), "a failed call wrote verdicts"
off = (np.arange(n + 1) * 40).astype(np.uint32)
arena = np.zeros(n * 40 + 64, np.uint8)
rc = L.fabgpu_sha256_p256_verify_batch(ctx.handle, n, p(arena), off.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)), p(b["qx"]), p(b["qy"]), p(b["r"]), p(b["s"]),
                                       bits.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)), p(st))
assert rc == %(want)d, rc
assert (bits == 0xA5A5A5A5A5A5A5A5).all() and (st == 0xEE).all()
if %(mode)r == "launch":
    # the provider surface: an infrastructure error, never "(false, nil)"
    import bccsp_sw_oracle as po
    csp = fabgpu.GPUCSP(device=0)
    k = csp.key_import((int.from_bytes(b["qx"][0].tobytes(), "big"), int.from_bytes(b["qy"][0].tobytes(), "big")))
    sig = po.marshal_ecdsa_signature(int.from_bytes(b["r"][0].tobytes(), "big"), int.from_bytes(b["s"][0].tobytes(), "big"))
    try:
        csp.verify(k, sig, b["e"][0].tobytes())
        raise SystemExit("verify returned a verdict while the device was failing")
    except fabgpu.FabgpuError:
        pass
# the block pass, on both routes (the second call finds the identities cached and takes the device walk): an error, never flags
import base64, json, os
blk = next(base64.b64decode(b_["block_b64"]) for b_ in json.load(open(%(root)r + "/tests/golden/ledger_blocks.json"))["blocks"] if b_["source"] == "v20" and b_["number"] == 6)
csp2 = fabgpu.GPUCSP(device=0)
csp2.set_option("pass_stage_min_bytes", 1)
for attempt in range(3):
    try:
        fabgpu.preverify_block(csp2, blk)
        raise SystemExit("the block pass answered while the device was failing")
    except fabgpu.FabgpuError:
        pass
print("FAULT_CONTRACT_OK")
"""


@pytest.mark.parametrize("mode,want", [("launch", -4), ("oom", -3)])
def test_device_faults_surface_as_infrastructure_errors_never_as_verdicts(mode, want):
    """FABGPU_FAULT_INJECT makes every kernel submission report hipErrorLaunchFailure ("launch") or every workspace allocation fail
    ("oom"): the C ABI must return FABGPU_ELAUNCH / FABGPU_ENOMEM, leave the caller's verdict arrays untouched, and the provider must
    raise an infrastructure error (the Go side then uses bccsp/sw) - never answer (false, nil)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, FABGPU_FAULT_INJECT=mode)
    r = subprocess.run([sys.executable, "-c", _FAULT_SNIPPET % {"root": root, "want": want, "mode": mode}], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0 and "FAULT_CONTRACT_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
