// +build fabgpu

// Package idemixgpu is the reference-side consumer of the block pass for idemix CREATOR signatures: a drop-in for
// handlers.NymVerifier (bccsp/idemix/handlers/nymsigner.go:58-95), wired where bccsp/idemix/bccsp.go:60-62 registers the
// verifier for *nymPublicKey:
//
//	base.AddWrapper(reflect.TypeOf(handlers.NewNymPublicKey(nil)), idemixgpu.New(&handlers.NymVerifier{...}, gpuProvider))
//
// It has no device call of its own (round 1 launched one kernel per Verify - slower than amcl on the CPU): the pseudonym
// signatures of a block are verified in ONE batched launch by the pass (bccsp/gpu PreVerifyBlock; an idemix MSP registers its
// issuer key with Provider.RegisterIdemixMSP), which memoises every verdict under (ipk.Hash of the issuer it verified under,
// Nym.x || Nym.y, signature bytes, SHA-256(message)).  Verify here recomputes that key from ITS arguments - the issuer hash from
// signerOpts.IssuerPK, so a verdict reached under another channel's issuer key of the same MSP id can never be found (ADVICE r2) -
// and looks it up; a hit that says "valid" is the only
// answer it gives itself - argument errors, misses, rejected proofs and every foreign type go to the reference's verifier,
// whose checks and error texts are therefore the reference's own.  Everything else of the idemix BCCSP (credentials,
// revocation, Signature.Ver and its pairings, signing) stays with bccsp/idemix.
// NOT compiled in this repository (no Go toolchain in the build image); Go 1.14 compatible.
package idemixgpu

import (
	"crypto/sha256"

	"github.com/hyperledger/fabric/bccsp"
	"github.com/hyperledger/fabric/bccsp/gpu"
	"github.com/hyperledger/fabric/bccsp/idemix/handlers"
)

// memo is the part of *gpu.Provider this verifier needs.
type memo interface {
	MemoLookupNym(issuerHash, nymX, nymY *[32]byte, signature, digest []byte) (status uint8, hit bool)
}

// NymVerifier answers from the verdict memo when it can and asks the wrapped software verifier otherwise.
type NymVerifier struct {
	SW   *handlers.NymVerifier // the reference's verifier: argument checks, error texts, and every tuple the memo does not know
	Memo memo
}

// issuerHash: idemix.IssuerPublicKey.Hash of the key the caller verifies under - what the device's issuer record carries too.  The
// reference's issuer key type returns exactly that as its SKI (bccsp/idemix/handlers/issuer.go:69-71, bridge/issuer.go Hash()).
func issuerHash(k bccsp.Key) (h [32]byte, ok bool) {
	ski := k.SKI()
	if len(ski) != 32 {
		return h, false
	}
	copy(h[:], ski)
	return h, true
}

// New wraps the reference's verifier.  provider == nil (the default BCCSP is not the GPU provider): plain software.
func New(sw *handlers.NymVerifier, provider *gpu.Provider) *NymVerifier {
	v := &NymVerifier{SW: sw}
	if provider != nil {
		v.Memo = provider
	}
	return v
}

// Verify: bccsp/idemix/handlers/nymsigner.go:62-95.  The reference's type assertions (*nymPublicKey, *issuerPublicKey are
// unexported) cannot be repeated here, so this method never ACCEPTS on the strength of its own checks: it only recognises the
// shape a memo entry can have (a 64-byte key, a non-empty signature) and otherwise delegates.  A memo hit proves that exactly
// this (pseudonym, signature, message) was verified under the very issuer key the caller names (the entry carries its ipk.Hash); the
// software verifier is still consulted for the issuer-key / option checks whenever the caller's opts are not the expected ones.
func (v *NymVerifier) Verify(k bccsp.Key, signature, digest []byte, opts bccsp.SignerOpts) (bool, error) {
	signerOpts, ok := opts.(*bccsp.IdemixNymSignerOpts)
	if v.Memo == nil || k == nil || !ok || signerOpts.IssuerPK == nil || len(signature) == 0 || len(digest) == 0 {
		return v.SW.Verify(k, signature, digest, opts) // the reference produces the error (or the verdict)
	}
	nymRaw, err := k.Bytes() // x || y (handlers/nym.go:89-91)
	if err != nil || len(nymRaw) != 64 {
		return v.SW.Verify(k, signature, digest, opts)
	}
	ih, ok := issuerHash(signerOpts.IssuerPK)
	if !ok {
		return v.SW.Verify(k, signature, digest, opts)
	}
	var qx, qy [32]byte
	copy(qx[:], nymRaw[:32])
	copy(qy[:], nymRaw[32:])
	d := sha256.Sum256(digest) // idemix hands the whole message over as "digest" (msp/idemixmsp.go:584-599)
	if st, hit := v.Memo.MemoLookupNym(&ih, &qx, &qy, signature, d[:]); hit && st == 0 {
		return true, nil
	}
	return v.SW.Verify(k, signature, digest, opts) // miss or rejected proof: the reference's answer and its error text
}
