// Cost-aware dealing of master blocks (deflate.c:916-923 is the unit: master blocks are independent).
//
// A master block of long runs of equal bytes costs several times a master block of text ("Per
// class"): contiguous equal-COUNT shards put a mixed corpus's expensive stretches on one or two ranks.  The shards
// are therefore balanced by an estimate of each master block's cost that is a function of the BYTES alone — every
// rank, and the in-process dealer, compute the same ranges from the same input; nothing is measured, nothing is
// exchanged.
#pragma once
#include <cstddef>
#include <vector>

namespace zamd {

// Cost of compressing in[begin, end) relative to the same number of bytes of text (1.0 per 1 000 000 bytes), from
// probes of 64 bytes every 1024.
double MasterBlockCost(const unsigned char* in, size_t begin, size_t end);

// first[s] .. first[s + 1]: the blocks of shard s — contiguous, in order, none empty while blocks >= shards, the
// largest shard's cost as small as a prefix walk makes it.  `first` gets shards + 1 entries.
void DealByCost(const std::vector<double>& cost, size_t shards, std::vector<size_t>* first);

}  // namespace zamd
