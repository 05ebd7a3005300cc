// +build fabgpu

// Benchmarks for SURVEY.md 8(d) "CPU baseline beside it, preferred form": bccsp/sw itself under `go test -bench`, on the box the GPU
// numbers come from, and the GPU provider through the same Go API.
//
//	go test -tags fabgpu -run xxx -bench . -benchtime 5s ./bccsp/gpu/
//
// BenchmarkSWVerify            BASELINE.json configs[0]: bccsp/sw ecdsa.Verify on CPU, 1k random P-256 signatures, single goroutine
//                              (bccsp/sw/impl.go:247-270 -> bccsp/sw/ecdsa.go:41-57; the provider is built exactly as the reference's own
//                              tests build it: bccsp/sw/impl_test.go:73-76)
// BenchmarkSWVerifyParallel    the same on every core (peer.validatorPoolSize defaults to NumCPU: core/peer/config.go:255-257)
// BenchmarkSWIdentityVerify    identity.Verify's two BCCSP calls per signature - Hash then Verify (msp/identities.go:169-196) - over the 1 856-byte
//                              endorsement messages of SURVEY 8(d)
// BenchmarkGPUVerifyOneByOne   the GPU provider's Verify called like bccsp/sw's, one signature per call, no block pass in front (memo misses:
//                              what an unpatched caller outside the validator sees - it should be NO faster than sw; it exists to show that)
// BenchmarkPreVerifyBlock      the block pass (PreVerifyBlock) over blocks of 1 000 transactions x 3 endorsements built with the reference's
//                              own protoutil helpers: validated tx/s = b.N x 1000 / elapsed (reported as tx/s), then every bccsp.Verify of the
//                              block answered from the memo (BenchmarkVerifyFromMemo)
//
// NOT run in this repository's CI: the build image has no Go toolchain (SURVEY.md 8(c)); bench.py's cpu_baseline is the OpenSSL proxy
// until someone runs these.  tools/check_go_sources.py keeps the file honest meanwhile (imports, C symbols, braces).
package gpu

import (
	"crypto/ecdsa"
	"crypto/elliptic"
	"crypto/rand"
	"crypto/sha256"
	"runtime"
	"sync/atomic"
	"testing"
	"time"

	"github.com/hyperledger/fabric/bccsp"
	"github.com/hyperledger/fabric/bccsp/sw"
	"github.com/hyperledger/fabric/bccsp/utils"
	"github.com/stretchr/testify/require"
)

type benchSig struct {
	key    bccsp.Key
	msg    []byte
	digest []byte
	sig    []byte
}

func swProvider(b testing.TB) bccsp.BCCSP {
	csp, err := sw.NewDefaultSecurityLevelWithKeystore(sw.NewDummyKeyStore())
	require.NoError(b, err)
	return csp
}

// n signatures, a fresh P-256 key pair each (SURVEY 8(d): "one fresh keypair per signature"), low-S DER, over msgLen random bytes
func benchSigs(b testing.TB, csp bccsp.BCCSP, n, msgLen int) []benchSig {
	out := make([]benchSig, n)
	for i := range out {
		priv, err := ecdsa.GenerateKey(elliptic.P256(), rand.Reader)
		require.NoError(b, err)
		k, err := csp.KeyImport(&priv.PublicKey, &bccsp.ECDSAGoPublicKeyImportOpts{Temporary: true})
		require.NoError(b, err)
		msg := make([]byte, msgLen)
		_, _ = rand.Read(msg)
		d := sha256.Sum256(msg)
		r, s, err := ecdsa.Sign(rand.Reader, priv, d[:])
		require.NoError(b, err)
		s, _ = utils.ToLowS(&priv.PublicKey, s)
		sig, err := utils.MarshalECDSASignature(r, s)
		require.NoError(b, err)
		out[i] = benchSig{key: k, msg: msg, digest: d[:], sig: sig}
	}
	return out
}

func BenchmarkSWVerify(b *testing.B) {
	csp := swProvider(b)
	sigs := benchSigs(b, csp, 1000, 32)
	b.ResetTimer()
	start := time.Now() // (testing.B.Elapsed is Go 1.20; the reference builds with Go 1.14)
	for i := 0; i < b.N; i++ {
		s := &sigs[i%len(sigs)]
		ok, err := csp.Verify(s.key, s.sig, s.digest, nil)
		if !ok || err != nil {
			b.Fatalf("bccsp/sw rejects a signature it made: %v", err)
		}
	}
	b.ReportMetric(float64(b.N)/time.Since(start).Seconds(), "verifies/s")
}

func BenchmarkSWVerifyParallel(b *testing.B) {
	csp := swProvider(b)
	sigs := benchSigs(b, csp, 1000, 32)
	var next uint64
	b.ReportMetric(float64(runtime.GOMAXPROCS(0)), "goroutines")
	b.ResetTimer()
	start := time.Now()
	b.RunParallel(func(pb *testing.PB) {
		for pb.Next() {
			s := &sigs[atomic.AddUint64(&next, 1)%uint64(len(sigs))]
			if ok, err := csp.Verify(s.key, s.sig, s.digest, nil); !ok || err != nil {
				b.Errorf("bccsp/sw rejects a signature it made: %v", err)
				return
			}
		}
	})
	b.ReportMetric(float64(b.N)/time.Since(start).Seconds(), "verifies/s")
}

func BenchmarkSWIdentityVerify(b *testing.B) {
	csp := swProvider(b)
	sigs := benchSigs(b, csp, 1000, 1856)
	b.SetBytes(1856)
	b.ResetTimer()
	for i := 0; i < b.N; i++ {
		s := &sigs[i%len(sigs)]
		d, err := csp.Hash(s.msg, &bccsp.SHA256Opts{})
		if err != nil {
			b.Fatal(err)
		}
		if ok, err := csp.Verify(s.key, s.sig, d, nil); !ok || err != nil {
			b.Fatalf("bccsp/sw rejects a signature it made: %v", err)
		}
	}
}

func gpuProvider(b testing.TB) bccsp.BCCSP {
	g, err := New(swProvider(b), Options{Devices: []int{0}, ConcurrentPasses: 2})
	if err != nil {
		b.Skipf("no MI355X here: %s", err)
	}
	return g
}

func BenchmarkGPUVerifyOneByOne(b *testing.B) {
	g := gpuProvider(b)
	defer g.(*Provider).Close()
	sigs := benchSigs(b, g, 1000, 32)
	b.ResetTimer()
	for i := 0; i < b.N; i++ {
		s := &sigs[i%len(sigs)]
		if ok, err := g.Verify(s.key, s.sig, s.digest, nil); !ok || err != nil {
			b.Fatalf("the GPU provider rejects a valid signature: %v", err)
		}
	}
}

func BenchmarkPreVerifyBlock(b *testing.B) {
	g := gpuProvider(b)
	p := g.(*Provider)
	defer p.Close()
	const nTx, nEnd = 1000, 3
	raw, tuples := buildSignedBlock(b, g, nTx, nEnd)
	require.Equal(b, nTx*(1+nEnd), len(tuples))
	// the provider meets the block's four identities (and builds their tables) outside the clock, as a peer does once per channel
	for k := 0; k < 3; k++ {
		_, err := p.PreVerifyBlock(raw, uint64(1000+k))
		require.NoError(b, err)
		p.EvictBlock(uint64(1000 + k))
	}
	b.SetBytes(int64(len(raw)))
	b.ResetTimer()
	start := time.Now()
	for i := 0; i < b.N; i++ {
		blk := append([]byte(nil), raw...) // a block a peer receives sits in memory nobody has seen before
		sum, err := p.PreVerifyBlock(blk, uint64(i+1))
		if err != nil {
			b.Fatal(err)
		}
		if len(sum.TxFlags) != nTx || sum.MemoSeeded != len(tuples) {
			b.Fatalf("pass over %d tx seeded %d memo entries for %d signatures", len(sum.TxFlags), sum.MemoSeeded, len(tuples))
		}
		p.EvictBlock(uint64(i + 1))
	}
	b.ReportMetric(float64(b.N)*nTx/time.Since(start).Seconds(), "tx/s")
}

func BenchmarkVerifyFromMemo(b *testing.B) {
	g := gpuProvider(b)
	p := g.(*Provider)
	defer p.Close()
	raw, tuples := buildSignedBlock(b, g, 1000, 3)
	_, err := p.PreVerifyBlock(raw, 42)
	require.NoError(b, err)
	digests := make([][]byte, len(tuples))
	for i, t := range tuples {
		d := sha256.Sum256(t.Msg)
		digests[i] = d[:]
	}
	b.ResetTimer()
	for i := 0; i < b.N; i++ {
		t := &tuples[i%len(tuples)]
		if ok, err := g.Verify(t.Key, t.Sig, digests[i%len(tuples)], nil); !ok || err != nil {
			b.Fatalf("memo answer for a valid signature: %v %v", ok, err)
		}
	}
	_, hits, _, _ := p.MemoStats()
	b.ReportMetric(float64(hits), "memo_hits")
}

// identity.Verify as the validators run it after a pass - Hash(msg) then Verify(k, sig, digest) per signature - with the digest memo
// answering Hash (round 6) and with bccsp/sw hashing (GPUOpts.NoHashMemo: round 5's provider).  ns/op is per signature.
func benchIdentityVerifyAfterPass(b *testing.B, noHashMemo bool) {
	ref, err := sw.NewDefaultSecurityLevelWithKeystore(sw.NewDummyKeyStore())
	require.NoError(b, err)
	g, err := New(ref, Options{Devices: []int{0}, NoHashMemo: noHashMemo})
	if err != nil {
		b.Skipf("no MI355X here: %s", err)
	}
	p := g.(*Provider)
	defer p.Close()
	raw, tuples := buildSignedBlock(b, g, 1000, 3)
	_, err = p.PreVerifyBlock(raw, 44)
	require.NoError(b, err)
	b.ResetTimer()
	b.RunParallel(func(pb *testing.PB) {
		i := 0
		for pb.Next() {
			t := &tuples[i%len(tuples)]
			i++
			digest, err := g.Hash(t.Msg, &bccsp.SHA256Opts{})
			if err != nil {
				b.Fatal(err)
			}
			if ok, err := g.Verify(t.Key, t.Sig, digest, nil); !ok || err != nil {
				b.Fatalf("memo answer for a valid signature: %v %v", ok, err)
			}
		}
	})
	hits, misses, _, _, _ := p.HashMemoStats()
	b.ReportMetric(float64(hits), "hash_memo_hits")
	b.ReportMetric(float64(misses), "hash_memo_misses")
}

func BenchmarkIdentityVerifyAfterPass(b *testing.B)           { benchIdentityVerifyAfterPass(b, false) }
func BenchmarkIdentityVerifyAfterPassNoHashMemo(b *testing.B) { benchIdentityVerifyAfterPass(b, true) }
