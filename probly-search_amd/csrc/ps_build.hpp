// ps_build.hpp — GPU bulk indexing: device half in ps_build.hip, host half
// (Index::bulk_load) in ps_index.cpp.
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>

namespace ps {

// A flat corpus after the device has tokenised it, grouped its tokens by term and reduced them to
// postings.  Terms are in an arbitrary (hash) order; `term_first_token` orders them by first occurrence.
struct GroupedCorpus {
  uint64_t n_tokens = 0;
  std::vector<uint32_t> term_pos, term_len;  // where the term's string is in the text
  std::vector<uint32_t> term_first_token;    // ordinal (text order) of the term's first token
  std::vector<uint32_t> term_post_begin;     // [n_terms + 1] postings of term t: [begin[t], begin[t+1])
  std::vector<uint32_t> post_doc;            // document ordinal (input order), ascending within a term
  std::vector<uint32_t> post_tf;             // [n_postings][F] DocumentPointer::term_frequency
  std::vector<uint32_t> field_length;        // [n_docs][F] DocumentDetails::field_length
};

// Tokenise (split on ' '), hash, sort by term, reduce to postings - all on `device`.  Returns false
// when two different terms collided in the 64-bit hash (detected, never silent): use the host indexer.
bool gpu_group_corpus(int device, uint32_t F, size_t n_docs, const char* text, const uint64_t* offsets, GroupedCorpus& out);

}  // namespace ps
