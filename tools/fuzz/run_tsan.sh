#!/bin/bash
# ThreadSanitizer over the host-side concurrency that has no device in it: the coalescer of one-signature calls (csrc/coalescer.h)
# with a fake device - 32..64 callers, launches that overlap arrivals, max_batch 2 / 4, no window.  Expect: no report, 0 wrong answers.
set -e
cd "$(dirname "$0")/../.."
g++ -O1 -g -std=c++17 -fsanitize=thread -Ifabric-mod_amd/csrc tools/fuzz/tsan_coalescer.cpp -o /tmp/tsan_coalescer -lpthread
/tmp/tsan_coalescer
# ... and the verdict memo's reader-writer lock (csrc/reader_lock.h): 12 readers checking an invariant that 3 writers break and restore
# under the write lock.  Expect: no report, invariant broken 0 times.
g++ -O1 -g -std=c++17 -fsanitize=thread -Ifabric-mod_amd/csrc tools/fuzz/tsan_reader_lock.cpp -o /tmp/tsan_reader_lock -lpthread
/tmp/tsan_reader_lock
