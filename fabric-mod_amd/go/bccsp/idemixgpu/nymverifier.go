// +build fabgpu

// Package idemixgpu is the reference-side binding of libfabgpu.so for idemix CREATOR signatures: a drop-in for
// handlers.NymVerifier (bccsp/idemix/handlers/nymsigner.go:58-95), wired where bccsp/idemix/bccsp.go:60-62 registers the
// verifier for *nymPublicKey.  Everything else of the idemix BCCSP (credentials, revocation, Signature.Ver and its pairings,
// signing) stays with bccsp/idemix.  NOT compiled in this repository (no Go toolchain in the build image): the C ABI it
// binds is exercised by tests/test_idemix_gpu.py through ctypes; fabric-mod_amd/csrc/idemix_host.cpp is the same logic in C++.
package idemixgpu

/*
#cgo CFLAGS: -I${SRCDIR}/../../../../include
#cgo LDFLAGS: -L${SRCDIR}/../../../lib -lfabgpu
#include <stdlib.h>
#include "fabgpu.h"
*/
import "C"

import (
	"sync"
	"unsafe"

	"github.com/golang/protobuf/proto"
	"github.com/hyperledger/fabric/bccsp"
	"github.com/hyperledger/fabric/bccsp/idemix/handlers"
	cryptolib "github.com/hyperledger/fabric/idemix"
	"github.com/pkg/errors"
)

// NymVerifier answers on the GPU when it can and asks the wrapped software verifier otherwise.
type NymVerifier struct {
	SW      *handlers.NymVerifier // the reference's verifier: fallback for every tuple the device does not decide
	Ctx     *C.fabgpu_ctx         // this verifier's own device context (cgo types are per package: not shared with bccsp/gpu)
	issuers sync.Map              // string(ipk bytes) -> int64 issuer id, -1 = not accelerated
}

// New is what bccsp/idemix/bccsp.go:60-62 calls under the fabgpu build tag; on any device error the caller keeps sw.
func New(sw *handlers.NymVerifier, device int) (*NymVerifier, error) {
	cfg := C.fabgpu_cfg{device: C.int32_t(device)}
	var ctx *C.fabgpu_ctx
	if rc := C.fabgpu_init(&cfg, &ctx); rc != 0 {
		return nil, errors.Errorf("Failed initializing GPU idemix verifier: %s", C.GoString(C.fabgpu_strerror(rc)))
	}
	return &NymVerifier{SW: sw, Ctx: ctx}, nil
}

func (v *NymVerifier) issuerID(ipk bccsp.Key) int64 {
	raw, err := ipk.Bytes() // marshalled idemix.IssuerPublicKey (handlers/issuer.go issuerPublicKey.Bytes)
	if err != nil {
		return -1
	}
	if id, ok := v.issuers.Load(string(raw)); ok {
		return id.(int64)
	}
	id := int64(-1)
	pk := &cryptolib.IssuerPublicKey{}
	if proto.Unmarshal(raw, pk) == nil && pk.HSk != nil && pk.HRand != nil &&
		len(pk.HSk.X) == 32 && len(pk.HSk.Y) == 32 && len(pk.HRand.X) == 32 && len(pk.HRand.Y) == 32 && len(pk.Hash) == 32 {
		var cid C.uint32_t
		if C.fabgpu_idemix_issuer_register(v.Ctx, u8(pk.HSk.X), u8(pk.HSk.Y), u8(pk.HRand.X), u8(pk.HRand.Y), u8(pk.Hash), &cid) == 0 {
			id = int64(cid)
		}
	}
	v.issuers.Store(string(raw), id)
	return id
}

func u8(b []byte) *C.uint8_t { return (*C.uint8_t)(unsafe.Pointer(&b[0])) }

// Verify keeps the argument checks and error text of handlers.NymVerifier.Verify (nymsigner.go:62-95).
func (v *NymVerifier) Verify(k bccsp.Key, signature, digest []byte, opts bccsp.SignerOpts) (bool, error) {
	signerOpts, ok := opts.(*bccsp.IdemixNymSignerOpts)
	if !ok || signerOpts.IssuerPK == nil || len(signature) == 0 {
		return v.SW.Verify(k, signature, digest, opts) // the reference produces the error
	}
	nymRaw, err := k.Bytes() // x || y (handlers/nym.go:89-91)
	sig := &cryptolib.NymSignature{}
	if err != nil || len(nymRaw) != 64 || proto.Unmarshal(signature, sig) != nil ||
		len(sig.ProofC) != 32 || len(sig.ProofSSk) != 32 || len(sig.ProofSRNym) != 32 || len(sig.Nonce) != 32 {
		return v.SW.Verify(k, signature, digest, opts)
	}
	id := v.issuerID(signerOpts.IssuerPK)
	if id < 0 {
		return v.SW.Verify(k, signature, digest, opts)
	}
	// one tuple per call here; the block-level pre-verify pass (bccsp/gpu PreVerifyBlock) batches the creators of a block
	off := [2]C.uint32_t{0, C.uint32_t(len(digest))}
	iss := C.uint32_t(id)
	var bits C.uint64_t
	var status C.uint8_t
	var arena *C.uint8_t
	if len(digest) > 0 {
		arena = u8(digest)
	}
	rc := C.fabgpu_idemix_nym_verify_batch(v.Ctx, 1, arena, &off[0], &iss, u8(nymRaw[:32]), u8(nymRaw[32:]),
		u8(sig.ProofC), u8(sig.ProofSSk), u8(sig.ProofSRNym), u8(sig.Nonce), &bits, &status)
	if rc != 0 || status == C.FABGPU_NYM_NEEDS_SW {
		return v.SW.Verify(k, signature, digest, opts) // infrastructure failure or a tuple outside the device's domain
	}
	if status == C.FABGPU_NYM_VALID {
		return true, nil
	}
	return false, errors.Errorf("pseudonym signature invalid: zero-knowledge proof is invalid") // idemix/nymsignature.go:105
}
