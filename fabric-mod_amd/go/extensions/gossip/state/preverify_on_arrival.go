// +build fabgpu

// The block pre-verify pass at block ARRIVAL (VERDICT r2 item 1, SURVEY.md 8(f) ranks 1 and 3).
//
// TrustBloc's fork already has the hook: GossipStateProviderExtension.AddPayload (extensions/gossip/state/state.go:31,75-77) wraps the
// function every block passes on its way into the payload buffer - blocks gossiped by other peers (gossip/state/state.go:328) and
// blocks the leader pulled from the orderer (gossip/state/state.go:785-787, called from internal/pkg/peer/blocksprovider).  What it
// receives is gossip.Payload{SeqNum, Data}, and Data IS proto.Marshal(block) (gossip/state/state.go:592 unmarshals it again when the
// committer gets there): the bytes the pass wants are in hand, nothing has to be marshalled a second time (the validator wrapper of
// round 2, extensions/validation/preverify.go, re-marshalled 50 MB per 10 000-transaction block inside Validate).
//
// So the pass runs HERE, on a goroutine of its own, while the block waits in the payload buffer for its turn at the committer:
//   - every creator / endorsement / orderer signature of the block and its TxID / proposal-hash digests in one device submission,
//     verdict memo seeded under memoSeq(channel, SeqNum);
//   - when the committer reaches the block, core/committer/txvalidator Validate finds the verdicts waiting: the wrapper in
//     extensions/validation/preverify.go asks HasBlock(seq) and skips its own marshal + pass; it evicts the entries when Validate returns;
//   - a block that never reaches Validate (dropped from the buffer, a duplicate) ages out: the memo holds at most 2^18 entries,
//     oldest block first (fabgpu_csp_memo_set_capacity).
// The pass is off the commit path altogether: validated tx/s is then bounded by max(pass, validation), not their sum.
//
// MCS.VerifyBlock (internal/peer/gossip/mcs.go:124-193) is NOT helped by this hook and is deliberately not hooked itself: both callers
// run it BEFORE AddPayload (gossip/gossip/channel verifyBlock; blocksprovider), it holds an unmarshalled *common.Block (a pass there
// would have to marshal 50 MB to save the one to three orderer signatures a block carries: ~0.1 ms of CPU), and its cost is
// protoutil.BlockDataHash - one SHA-256 over the whole BlockData, 34 ms for 50 MB on the bench host - which no GPU lane can run faster
// than the CPU's SHA extensions (DESIGN.md section 9).  The orderers' signatures are still tuples of the pass (kind 2): the
// BlockValidation policy evaluated again at commit time (core/committer/txvalidator, config blocks) finds them in the memo.
//
// Apply in extensions/gossip/state/state.go:
//	:58-60   NewGossipStateProviderExtension keeps  arrival: newArrivalPasses(chainID)  in gossipStateProviderExtension
//	:75-77   AddPayload:
//	-	return handle
//	+	return s.arrival.wrap(handle)
// The reference builds the AddPayload wrapper anew for EVERY payload (gossip/state/state.go:328 and :787 call
// s.extension.AddPayload(s.addPayload)(payload, ...)), so whatever must outlive one payload - the limiter on passes in flight, the
// provider lookup - lives in arrivalPasses, made once per channel (ADVICE r3: a limiter created inside the wrapper limited nothing).
// NOT compiled in this repository (no Go toolchain in the build image); Go 1.14 compatible.
package state

import (
	proto "github.com/hyperledger/fabric-protos-go/gossip"
	"github.com/hyperledger/fabric/bccsp/factory"
	"github.com/hyperledger/fabric/bccsp/gpu"
	"github.com/hyperledger/fabric/common/flogging"
)

var arrivalLogger = flogging.MustGetLogger("extensions.gossip.state.preverify")

// MemoSeq names a block in the verdict memo: channels share one provider, block numbers repeat across channels.  The validator
// wrapper (extensions/validation/preverify.go) computes the same name.
func MemoSeq(channelID string, number uint64) uint64 {
	h := uint64(14695981039346656037) // FNV-1a of the channel id, folded over the block number: a name, not a security boundary
	for i := 0; i < len(channelID); i++ {
		h ^= uint64(channelID[i])
		h *= 1099511628211
	}
	return h ^ (number * 0x9E3779B97F4A7C15)
}

// at most this many passes in flight per channel: a burst of arriving blocks (state transfer) must not queue device work without bound
const maxArrivalPasses = 4

// arrivalPasses is the per-channel state of the hook: ONE limiter and ONE provider lookup for the life of the channel.
type arrivalPasses struct {
	channelID string
	pre       gpu.BlockPreVerifier // nil: the default BCCSP cannot pre-verify
	slots     chan struct{}        // at most maxArrivalPasses passes in flight for this channel
}

func newArrivalPasses(channelID string) *arrivalPasses {
	pre, _ := factory.GetDefault().(gpu.BlockPreVerifier)
	return &arrivalPasses{channelID: channelID, pre: pre, slots: make(chan struct{}, maxArrivalPasses)}
}

// wrap is what AddPayload returns: handle first - it decides whether the block is kept at all (gossip/state/state.go:804 drops blocks
// below the ledger height or beyond the buffer window, payloads.Push drops duplicates) - and only a block that was KEPT is pre-verified:
// a pass for a block that never reaches Validate would seed memo entries nobody evicts, and the bounded memo drops its OLDEST block
// first - the one the committer needs next (ADVICE r3).
func (a *arrivalPasses) wrap(handle func(payload *proto.Payload, blockingMode bool) error) func(payload *proto.Payload, blockingMode bool) error {
	if a == nil || a.pre == nil {
		return handle
	}
	return func(payload *proto.Payload, blockingMode bool) error {
		err := handle(payload, blockingMode)
		if err != nil || payload == nil || len(payload.Data) == 0 {
			return err
		}
		seq := MemoSeq(a.channelID, payload.SeqNum)
		select {
		case a.slots <- struct{}{}:
			data := payload.Data // not retained past the pass: the provider copies what it keeps (cgo pointer rules)
			num := payload.SeqNum
			go func() {
				defer func() { <-a.slots }()
				if a.pre.HasBlock(seq) {
					return // a duplicate of a block that is already waiting
				}
				if _, err := a.pre.PreVerifyBlock(data, seq); err != nil {
					arrivalLogger.Debugf("[%s] block %d: pre-verify at arrival failed (%s); the validators will use bccsp/sw", a.channelID, num, err)
				}
			}()
		default: // enough passes in flight: this block is pre-verified at Validate (or not at all: bccsp/sw is always right)
		}
		return nil
	}
}
