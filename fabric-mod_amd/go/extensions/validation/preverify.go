// +build fabgpu

// preVerifying is the second (batch) plug point of SURVEY.md 8(b): TrustBloc's extension hook extensions/validation.NewTxValidator
// (extensions/validation/validation.go:48-64) returns the channel's txvalidator.Validator; this file wraps it.  Before the
// unchanged v14 / v20 validator fans a block out to its goroutines (core/committer/txvalidator/v20/validator.go:182-267), the
// whole block - marshalled once - goes to the GPU provider: every creator signature (core/common/validation/msgvalidation.go:
// 258-298), every endorsement signature (core/common/validation/statebased/validator_keylevel.go:246-258), the orderers' block
// signatures (internal/peer/gossip/mcs.go:166-193) and the TxID / proposal-hash digests in ONE device submission, which seeds the
// provider's verdict memo.  The validators then run exactly as before; their bccsp.Verify calls hit the memo.  When Validate
// returns the block's memo entries are evicted.
//
// Since round 3 the pass normally ran long before: extensions/gossip/state/preverify_on_arrival.go submits the block when it ARRIVES
// (its marshalled bytes are in hand there).  Validate asks the provider whether the block's verdicts are waiting (HasBlock) and only
// marshals and submits the block itself when they are not (a block that came in while the arrival hook's slots were taken, a peer
// built without that hook).
//
// Nothing here is consensus input: the pass only pre-answers bccsp.Verify for byte strings the validators themselves present
// (see bccsp/gpu/gpu.go); if the provider is not the GPU one, or the pass fails, validation proceeds on bccsp/sw unchanged.
//
// Apply in extensions/validation/validation.go:58-63:
//	-	return &txvalidator.ValidationRouter{...}
//	+	return newPreVerifying(&txvalidator.ValidationRouter{...}, cryptoProvider, channelID)
// NOT compiled in this repository (no Go toolchain in the build image); Go 1.14 compatible.
package validation

import (
	"github.com/golang/protobuf/proto"
	"github.com/hyperledger/fabric-protos-go/common"
	"github.com/hyperledger/fabric/bccsp"
	"github.com/hyperledger/fabric/bccsp/gpu"
	"github.com/hyperledger/fabric/common/flogging"
	"github.com/hyperledger/fabric/core/committer/txvalidator"
)

var preLogger = flogging.MustGetLogger("extensions.validation.preverify")

type preVerifying struct {
	next      txvalidator.Validator
	pre       gpu.BlockPreVerifier // nil: the default BCCSP cannot pre-verify
	channelID string
}

func newPreVerifying(next txvalidator.Validator, cryptoProvider bccsp.BCCSP, channelID string) txvalidator.Validator {
	pre, _ := cryptoProvider.(gpu.BlockPreVerifier)
	if pre == nil {
		return next
	}
	return &preVerifying{next: next, pre: pre, channelID: channelID}
}

// memoSeq names the block in the memo: channels share one provider, block numbers repeat across channels.  (Same function as
// extensions/gossip/state.MemoSeq - repeated here so that this file stands alone in a build without the arrival hook.)
func memoSeq(channelID string, number uint64) uint64 {
	h := uint64(14695981039346656037) // FNV-1a of the channel id, folded over the block number: a name, not a security boundary
	for i := 0; i < len(channelID); i++ {
		h ^= uint64(channelID[i])
		h *= 1099511628211
	}
	return h ^ (number * 0x9E3779B97F4A7C15)
}

// Validate implements txvalidator.Validator (core/committer/txvalidator/router.go:17-22).
func (v *preVerifying) Validate(block *common.Block) error {
	if block == nil || block.Header == nil || block.Data == nil || len(block.Data.Data) == 0 {
		return v.next.Validate(block)
	}
	seq := memoSeq(v.channelID, block.Header.Number)
	if v.pre.HasBlock(seq) { // pre-verified when it arrived: nothing to marshal, nothing to submit
		defer v.pre.EvictBlock(seq)
		return v.next.Validate(block)
	}
	raw, err := proto.Marshal(block) // one pass over the block's bytes; the pass walks them in place
	if err != nil {
		return v.next.Validate(block)
	}
	sum, err := v.pre.PreVerifyBlock(raw, seq)
	if err != nil {
		preLogger.Warningf("[%s] block %d: GPU pre-verify pass failed (%s); validating on bccsp/sw", v.channelID, block.Header.Number, err)
		return v.next.Validate(block)
	}
	defer v.pre.EvictBlock(seq)
	preLogger.Debugf("[%s] block %d: %d signatures pre-verified (%d orderer), %d memo entries", v.channelID, block.Header.Number,
		sum.Tuples, sum.BlockSigs, sum.MemoSeeded)
	return v.next.Validate(block)
}
