// +build fabgpu

// Differential tests of the GPU provider against bccsp/sw, written after the reference's own tests for this path
// (bccsp/sw/impl_test.go:525-586, 931-964, 1008-1032; bccsp/sw/ecdsa_test.go:47-73; bccsp/utils/ecdsa_test.go:20-88;
// msp/msp_test.go:494-536).  They run wherever a Go toolchain and an MI355X exist - neither is in the build image of this
// repository, where the same cases are exercised through the C ABI by tests/test_gpu_parity.py against the CPU oracle.
package gpu

import (
	"crypto/ecdsa"
	"crypto/elliptic"
	"crypto/rand"
	"crypto/sha256"
	"math/big"
	"testing"

	"github.com/hyperledger/fabric/bccsp"
	"github.com/hyperledger/fabric/bccsp/sw"
	"github.com/hyperledger/fabric/bccsp/utils"
	"github.com/stretchr/testify/require"
)

func providers(t *testing.T) (bccsp.BCCSP, bccsp.BCCSP) {
	ref, err := sw.NewDefaultSecurityLevelWithKeystore(sw.NewDummyKeyStore())
	require.NoError(t, err)
	g, err := New(ref, Options{Devices: []int{0}})
	if err != nil {
		t.Skipf("no MI355X here: %s", err)
	}
	return g, ref
}

// same (valid, err == nil, error text) from both providers
func same(t *testing.T, g, ref bccsp.BCCSP, gk, rk bccsp.Key, sig, digest []byte) {
	v1, e1 := g.Verify(gk, sig, digest, nil)
	v2, e2 := ref.Verify(rk, sig, digest, nil)
	require.Equal(t, v2, v1)
	require.Equal(t, e2 == nil, e1 == nil)
	if e2 != nil {
		require.Equal(t, e2.Error(), e1.Error())
	}
}

func importBoth(t *testing.T, g, ref bccsp.BCCSP, pub *ecdsa.PublicKey) (bccsp.Key, bccsp.Key) {
	gk, err := g.KeyImport(pub, &bccsp.ECDSAGoPublicKeyImportOpts{Temporary: true})
	require.NoError(t, err)
	rk, err := ref.KeyImport(pub, &bccsp.ECDSAGoPublicKeyImportOpts{Temporary: true})
	require.NoError(t, err)
	return gk, rk
}

func TestSignVerifyTamperLikeTheReference(t *testing.T) {
	g, ref := providers(t)
	for i := 0; i < 200; i++ {
		priv, _ := ecdsa.GenerateKey(elliptic.P256(), rand.Reader)
		gk, rk := importBoth(t, g, ref, &priv.PublicKey)
		msg := make([]byte, 1+i*7)
		rand.Read(msg)
		digest := sha256.Sum256(msg)
		r, s, _ := ecdsa.Sign(rand.Reader, priv, digest[:])
		s, _ = utils.ToLowS(&priv.PublicKey, s)
		sig, _ := utils.MarshalECDSASignature(r, s)
		same(t, g, ref, gk, rk, sig, digest[:])
		digest[3] ^= 0x40 // tampered digest
		same(t, g, ref, gk, rk, sig, digest[:])
		digest[3] ^= 0x40
		hs := new(big.Int).Sub(elliptic.P256().Params().N, s) // high-S twin: (false, error) in bccsp/sw
		high, _ := utils.MarshalECDSASignature(r, hs)
		same(t, g, ref, gk, rk, high, digest[:])
		r1, _ := utils.MarshalECDSASignature(new(big.Int).Add(r, big.NewInt(1)), s)
		same(t, g, ref, gk, rk, r1, digest[:])
	}
}

func TestLowSBoundaryAndDegenerateValues(t *testing.T) {
	g, ref := providers(t)
	priv, _ := ecdsa.GenerateKey(elliptic.P256(), rand.Reader)
	gk, rk := importBoth(t, g, ref, &priv.PublicKey)
	digest := sha256.Sum256([]byte("boundary"))
	half := utils.GetCurveHalfOrdersAt(elliptic.P256())
	for _, s := range []*big.Int{half, new(big.Int).Add(half, big.NewInt(1)), big.NewInt(0), big.NewInt(-1), big.NewInt(1)} {
		for _, r := range []*big.Int{big.NewInt(1), big.NewInt(0), big.NewInt(-1), elliptic.P256().Params().N} {
			sig, err := utils.MarshalECDSASignature(r, s)
			if err != nil {
				continue
			}
			same(t, g, ref, gk, rk, sig, digest[:])
		}
	}
}

func TestDERNegativesOfImplTest(t *testing.T) {
	g, ref := providers(t)
	priv, _ := ecdsa.GenerateKey(elliptic.P256(), rand.Reader)
	gk, rk := importBoth(t, g, ref, &priv.PublicKey)
	digest := sha256.Sum256([]byte("der"))
	for _, sig := range [][]byte{
		nil, {}, {0x30}, {0x30, 0x00}, {0x30, 0x03, 0x02, 0x01}, {0x30, 0x06, 0x02, 0x01, 0x01, 0x02, 0x01, 0x01, 0x00}, // trailing byte is ignored by asn1.Unmarshal
		{0x30, 0x07, 0x02, 0x02, 0x00, 0x01, 0x02, 0x01, 0x01},                                                           // non-minimal integer
		{0x30, 0x81, 0x06, 0x02, 0x01, 0x01, 0x02, 0x01, 0x01},                                                           // long-form length where short would do
	} {
		same(t, g, ref, gk, rk, sig, digest[:])
	}
	same(t, g, ref, nil, nil, []byte{0x30, 0x00}, digest[:]) // nil key
	same(t, g, ref, gk, rk, []byte{0x30, 0x06, 0x02, 0x01, 0x01, 0x02, 0x01, 0x01}, nil)
}

// A key of another curve must flow through untouched (ADVICE r1: a P-384 endorser panicked the round-1 provider).
func TestForeignCurvesAndNilKeysGoToSW(t *testing.T) {
	g, ref := providers(t)
	priv, _ := ecdsa.GenerateKey(elliptic.P384(), rand.Reader)
	gk, rk := importBoth(t, g, ref, &priv.PublicKey)
	digest := sha256.Sum256([]byte("p384"))
	r, s, _ := ecdsa.Sign(rand.Reader, priv, digest[:])
	s, _ = utils.ToLowS(&priv.PublicKey, s)
	sig, _ := utils.MarshalECDSASignature(r, s)
	same(t, g, ref, gk, rk, sig, digest[:])
	same(t, g, ref, nil, nil, sig, digest[:])
}

// Every key this provider hands out is bccsp/sw's own key object (VERDICT r2: a wrapper type broke KeyDeriv on imported public
// keys - bccsp/sw dispatches on reflect.TypeOf(key), impl.go:110, and registers a KeyDeriver for *ecdsaPublicKey, new.go:86).
func TestImportedKeysKeepEveryVerbOfBCCSPSW(t *testing.T) {
	g, ref := providers(t)
	priv, _ := ecdsa.GenerateKey(elliptic.P256(), rand.Reader)
	gk, rk := importBoth(t, g, ref, &priv.PublicKey)
	require.IsType(t, rk, gk) // the very type bccsp/sw returns
	// KeyDeriv on an imported public key (bccsp/sw/keyderiv.go: ECDSAReRandKeyOpts on *ecdsaPublicKey)
	opts := &bccsp.ECDSAReRandKeyOpts{Temporary: true, Expansion: []byte{1, 2, 3}}
	d1, e1 := g.KeyDeriv(gk, opts)
	d2, e2 := ref.KeyDeriv(rk, opts)
	require.NoError(t, e2)
	require.NoError(t, e1)
	require.Equal(t, d2.SKI(), d1.SKI())
	// a key generated and stored by the provider: GetKey(ski) round-trips, Sign works, Verify of that signature too
	ks, err := sw.NewFileBasedKeyStore(nil, t.TempDir(), false)
	require.NoError(t, err)
	ref2, err := sw.NewDefaultSecurityLevelWithKeystore(ks)
	require.NoError(t, err)
	g2, err := New(ref2, Options{Devices: []int{0, 0}}) // two contexts on the one device: the pool behind one provider
	require.NoError(t, err)
	k, err := g2.KeyGen(&bccsp.ECDSAP256KeyGenOpts{Temporary: false})
	require.NoError(t, err)
	back, err := g2.GetKey(k.SKI())
	require.NoError(t, err)
	require.Equal(t, k.SKI(), back.SKI())
	digest := sha256.Sum256([]byte("own key"))
	sig, err := g2.Sign(back, digest[:], nil)
	require.NoError(t, err)
	pub, err := back.PublicKey()
	require.NoError(t, err)
	ok, err := g2.Verify(pub, sig, digest[:], nil)
	require.NoError(t, err)
	require.True(t, ok)
	ok, err = g2.Verify(back, sig, digest[:], nil) // bccsp/sw verifies with a private key's public half (bccsp/sw/ecdsa.go:59-69)
	require.NoError(t, err)
	require.True(t, ok)
}

// The pass seeds the memo from the BYTES of a marshalled block; the per-signature calls the validators make afterwards hit it.
// (Block construction: the reference's own protoutil helpers; the C++ walker is pinned against the reference's sample ledgers
// by tests/test_ledger_goldens.py.)
func TestPreVerifyBlockSeedsTheMemoAndEvicts(t *testing.T) {
	g, _ := providers(t)
	p := g.(*Provider)
	blockBytes, tuples := buildSignedBlock(t, g, 50, 3) // helper in block_helper_test.go (uses protoutil.CreateSignedTx)
	sum, err := p.PreVerifyBlock(blockBytes, 42)
	require.NoError(t, err)
	require.Equal(t, len(tuples), sum.MemoSeeded)
	_, hits0, _, _ := p.MemoStats()
	for _, tu := range tuples {
		digest := sha256.Sum256(tu.Msg)
		ok, err := g.Verify(tu.Key, tu.Sig, digest[:], nil)
		require.NoError(t, err)
		require.True(t, ok)
	}
	_, hits1, _, _ := p.MemoStats()
	require.Equal(t, uint64(len(tuples)), hits1-hits0)
	p.EvictBlock(42)
	entries, _, _, _ := p.MemoStats()
	require.Equal(t, uint64(0), entries)
}

// identity.Verify hashes before it verifies (msp/identities.go:173-181).  After a pass, Hash(msg, &bccsp.SHA256Opts{}) for a message of
// the block comes from the digest memo - and equals bccsp/sw's digest, because a hit is byte equality with what the device hashed; one
// flipped byte, another length, other HashOpts, an evicted block: bccsp/sw's answer (same digest by definition, counted as a miss).
func TestHashFromTheDigestMemo(t *testing.T) {
	g, ref := providers(t)
	p := g.(*Provider)
	blockBytes, tuples := buildSignedBlock(t, g, 50, 3)
	_, err := p.PreVerifyBlock(blockBytes, 43)
	require.NoError(t, err)
	hits0, _, held, _, _ := p.HashMemoStats()
	require.Equal(t, uint64(1), held)
	for _, tu := range tuples {
		want, err := ref.Hash(tu.Msg, &bccsp.SHA256Opts{})
		require.NoError(t, err)
		got, err := g.Hash(tu.Msg, &bccsp.SHA256Opts{})
		require.NoError(t, err)
		require.Equal(t, want, got)
		ok, err := g.Verify(tu.Key, tu.Sig, got, nil) // Hash then Verify, as identity.Verify does
		require.NoError(t, err)
		require.True(t, ok)
	}
	hits1, miss1, _, _, _ := p.HashMemoStats()
	require.Equal(t, uint64(len(tuples)), hits1-hits0)
	// anything but the block's own bytes is hashed by bccsp/sw - with bccsp/sw's result
	for i, tu := range tuples {
		bad := append([]byte(nil), tu.Msg...)
		bad[(i*131)%len(bad)] ^= 0x20
		want, _ := ref.Hash(bad, &bccsp.SHA256Opts{})
		got, err := g.Hash(bad, &bccsp.SHA256Opts{})
		require.NoError(t, err)
		require.Equal(t, want, got)
		want384, _ := ref.Hash(tu.Msg, &bccsp.SHA384Opts{}) // other HashOpts never ask the memo
		got384, err := g.Hash(tu.Msg, &bccsp.SHA384Opts{})
		require.NoError(t, err)
		require.Equal(t, want384, got384)
	}
	_, errNil := g.Hash(tuples[0].Msg, nil) // bccsp/sw's error for nil opts (bccsp/sw/impl.go:179-181)
	_, errRef := ref.Hash(tuples[0].Msg, nil)
	require.Error(t, errNil)
	require.Equal(t, errRef.Error(), errNil.Error())
	_, miss2, _, _, _ := p.HashMemoStats()
	require.Equal(t, uint64(len(tuples)), miss2-miss1) // the flipped messages (SHA384 and nil opts never reached the library)
	p.EvictBlock(43)
	_, _, held, _, _ = p.HashMemoStats()
	require.Equal(t, uint64(0), held)
	got, err := g.Hash(tuples[0].Msg, &bccsp.SHA256Opts{})
	require.NoError(t, err)
	want, _ := ref.Hash(tuples[0].Msg, &bccsp.SHA256Opts{})
	require.Equal(t, want, got)
}
