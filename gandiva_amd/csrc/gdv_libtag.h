// Kernel identity vs the device function library (round 3).
//
// A fused kernel is its generated text AND the device functions that text reaches in
// gdv_device_lib.hpp.  Rounds 1-2 hashed the whole header into every kernel name, so adding a
// string function renamed the float64 projection kernel and invalidated every profile taken on it.
// LibraryIndex splits the header into top-level items (one #define, struct, typedef or function
// each), records which identifiers an item mentions, and hashes — for one generated text — only
// the items reachable from it by name, plus a small base of items that cannot be attributed to a
// name (conditional-compilation lines, the token-pasting macro families and their instantiations).
// Comments and whitespace are not part of the hash.
//
// The tag names a kernel (evidence, in-memory cache); the on-disk code-object cache is additionally
// keyed by the hash of the WHOLE header (FullTag), so a stale object can never be loaded even if
// the reachability analysis missed an edge.
#pragma once

#include <cstdint>
#include <map>
#include <string>
#include <vector>

namespace gdv {

class LibraryIndex {
 public:
  explicit LibraryIndex(const std::string& header_source);

  // hash of the items of the library that `kernel_text` reaches (decimal string)
  std::string TagFor(const std::string& kernel_text) const;
  // hash of the whole header, comments included
  const std::string& FullTag() const { return full_tag_; }

  struct Item {
    std::string name;                // "" for base items (always hashed)
    std::string code;                // comments stripped, whitespace collapsed
    std::vector<std::string> idents; // identifiers mentioned (deduplicated)
  };
  const std::vector<Item>& items() const { return items_; }
  // names of the items TagFor would hash for this text (diagnostics / tests)
  std::vector<std::string> ReachedFrom(const std::string& kernel_text) const;

  // the library embedded in libgandiva_amd.so
  static const LibraryIndex& Embedded();

 private:
  std::vector<bool> Reach(const std::string& kernel_text) const;
  std::vector<Item> items_;
  std::multimap<std::string, size_t> by_name_;
  std::string full_tag_;
};

uint64_t Fnv1a64(const std::string& s);

}  // namespace gdv
