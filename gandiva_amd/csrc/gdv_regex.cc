#include "gdv_regex.h"

#include <algorithm>
#include <array>
#include <bitset>
#include <cctype>
#include <cstring>
#include <initializer_list>
#include <memory>
#include <vector>

namespace gdv {
namespace {

using ByteSet = std::bitset<256>;

// syntax tree: leaves are byte sets (one automaton position each)
struct Re {
  enum Kind { kEmpty, kSet, kCat, kAlt, kStar, kPlus, kOpt, kAssert } kind = kEmpty;
  int cond = 0;  // kAssert: 1 word boundary, 2 not a word boundary, 3 start of the text, 4 end of the text
  ByteSet set;
  std::unique_ptr<Re> a, b;
  int pos = -1;
};
using ReP = std::unique_ptr<Re>;

ReP Mk(Re::Kind k, ReP a = nullptr, ReP b = nullptr) {
  ReP r(new Re);
  r->kind = k;
  r->a = std::move(a);
  r->b = std::move(b);
  return r;
}
ReP MkSet(const ByteSet& s) {
  ReP r(new Re);
  r->kind = Re::kSet;
  r->set = s;
  return r;
}
ReP MkAssert(int cond) {
  ReP r(new Re);
  r->kind = Re::kAssert;
  r->cond = cond;
  return r;
}
ReP Clone(const Re& x) {
  ReP r(new Re);
  r->kind = x.kind;
  r->set = x.set;
  r->cond = x.cond;
  if (x.a) r->a = Clone(*x.a);
  if (x.b) r->b = Clone(*x.b);
  return r;
}
int Leaves(const Re& x) { return x.kind == Re::kSet ? 1 : (x.a ? Leaves(*x.a) : 0) + (x.b ? Leaves(*x.b) : 0); }

ByteSet Range(int lo, int hi) {
  ByteSet s;
  for (int c = lo; c <= hi; c++) s.set(static_cast<size_t>(c));
  return s;
}
const ByteSet& ContinuationBytes() {
  static const ByteSet s = Range(0x80, 0xBF);
  return s;
}
// one whole UTF-8 character whose FIRST byte is in `lead`: the lead byte, then its continuation bytes
ReP WholeCharacter(const ByteSet& lead) { return Mk(Re::kCat, MkSet(lead), Mk(Re::kStar, MkSet(ContinuationBytes()))); }
// bytes that can start a character: ASCII and the lead bytes of multi-byte sequences
ByteSet AnyLead() { return Range(0x00, 0x7F) | Range(0xC2, 0xF4); }

ByteSet Digits() { return Range('0', '9'); }
ByteSet WordChars() { return Range('0', '9') | Range('a', 'z') | Range('A', 'Z') | Range('_', '_'); }
// [[:space:]] holds the vertical tab, Perl's \s / \S do not (RE2: \s = [\t\n\f\r ])
ByteSet Spaces() {
  ByteSet s;
  for (char c : {' ', '\t', '\n', '\r', '\f', '\v'}) s.set(static_cast<unsigned char>(c));
  return s;
}
ByteSet PerlSpaces() {
  ByteSet s = Spaces();
  s.reset('\v');
  return s;
}

// (?i): a set that holds an ASCII letter holds it in both cases
ByteSet FoldCase(ByteSet s) {
  for (int c = 'a'; c <= 'z'; c++)
    if (s.test(static_cast<size_t>(c)) || s.test(static_cast<size_t>(c - 32))) { s.set(static_cast<size_t>(c)); s.set(static_cast<size_t>(c - 32)); }
  return s;
}

ReP Bytes(std::initializer_list<int> bytes) {
  ReP seq;
  for (int k : bytes) seq = seq ? Mk(Re::kCat, std::move(seq), MkSet(Range(k, k))) : MkSet(Range(k, k));
  return seq;
}
// RE2's (?i) is Unicode simple folding.  Two characters outside ASCII fold onto ASCII letters: U+212A KELVIN SIGN (E2 84 AA)
// onto k, U+017F LATIN SMALL LETTER LONG S (C5 BF) onto s.  A folded set that holds the letter matches the character too ...
ReP FoldedSet(const ByteSet& folded) {
  ReP r = MkSet(folded);
  if (folded.test('k')) r = Mk(Re::kAlt, std::move(r), Bytes({0xE2, 0x84, 0xAA}));
  if (folded.test('s')) r = Mk(Re::kAlt, std::move(r), Bytes({0xC5, 0xBF}));
  return r;
}
// ... and "one character that is NOT in the (folded) set" leaves the two out when the set holds their letters
ReP WholeCharacterOutside(const ByteSet& excluded, bool fold) {
  ByteSet lead = AnyLead() & ~excluded;
  const bool kelvin = fold && excluded.test('k'), long_s = fold && excluded.test('s');
  if (kelvin) lead.reset(0xE2);
  if (long_s) lead.reset(0xC5);
  ReP r = WholeCharacter(lead);
  auto tail = [] { return Mk(Re::kStar, MkSet(ContinuationBytes())); };
  auto cont_but = [](int b) { ByteSet s = ContinuationBytes(); s.reset(static_cast<size_t>(b)); return MkSet(s); };
  if (kelvin) {
    r = Mk(Re::kAlt, std::move(r), Mk(Re::kCat, Mk(Re::kCat, Bytes({0xE2}), cont_but(0x84)), tail()));
    r = Mk(Re::kAlt, std::move(r), Mk(Re::kCat, Mk(Re::kCat, Bytes({0xE2, 0x84}), cont_but(0xAA)), tail()));
  }
  if (long_s) r = Mk(Re::kAlt, std::move(r), Mk(Re::kCat, Mk(Re::kCat, Bytes({0xC5}), cont_but(0xBF)), tail()));
  return r;
}

// Bounds on what the parser will build (patterns arrive from user SQL): the automaton has 63 positions, so a longer tree can
// only end in the "more than 63 positions" error — say so before building it.  Nesting and the number of atoms are capped too:
// the tree is walked recursively (Build, Clone, the destructors), empty groups and assertions add depth without positions.
constexpr size_t kMaxPatternBytes = 4096;
constexpr int kMaxNesting = 64;
constexpr int kMaxAtoms = 512;

class Parser {
 public:
  Parser(const std::string& p, bool fold, bool dot_nl) : p_(p), fold_(fold), dot_nl_(dot_nl) {}
  Status Parse(ReP* out) {
    if (p_.size() > kMaxPatternBytes)
      return Status::CodeGenError("regular expression of " + std::to_string(p_.size()) + " bytes not supported by the HIP backend: longer than " +
                                  std::to_string(kMaxPatternBytes) + " bytes");
    GDV_RETURN_NOT_OK(Alt(out));
    if (i_ < p_.size()) return Bad(p_[i_] == ')' ? "unmatched ')'" : "unexpected character");
    return Status::OK();
  }

 private:
  Status Bad(const std::string& why) {
    return Status::CodeGenError("regular expression '" + p_ + "' not supported yet by the HIP backend: " + why + " (at offset " +
                                std::to_string(i_) + ")");
  }
  bool More() const { return i_ < p_.size(); }
  Status TooManyPositions() { return Bad("more than 63 automaton positions"); }
  Status Alt(ReP* out) {
    ReP left;
    GDV_RETURN_NOT_OK(Cat(&left));
    int leaves = Leaves(*left);
    while (More() && p_[i_] == '|') {
      i_++;
      ReP right;
      GDV_RETURN_NOT_OK(Cat(&right));
      if ((leaves += Leaves(*right)) > 63) return TooManyPositions();
      left = Mk(Re::kAlt, std::move(left), std::move(right));
    }
    *out = std::move(left);
    return Status::OK();
  }
  Status Cat(ReP* out) {
    ReP left = Mk(Re::kEmpty);
    int leaves = 0;
    while (More() && p_[i_] != '|' && p_[i_] != ')') {
      ReP piece;
      GDV_RETURN_NOT_OK(Repeat(&piece));
      if ((leaves += Leaves(*piece)) > 63) return TooManyPositions();
      left = left->kind == Re::kEmpty ? std::move(piece) : Mk(Re::kCat, std::move(left), std::move(piece));
    }
    *out = std::move(left);
    return Status::OK();
  }
  Status Repeat(ReP* out) {
    ReP atom;
    GDV_RETURN_NOT_OK(Atom(&atom));
    while (More()) {
      const char c = p_[i_];
      if (c == '*' || c == '+' || c == '?') {
        i_++;
        if (++atoms_ > kMaxAtoms) return Bad("more than " + std::to_string(kMaxAtoms) + " atoms and quantifiers");
        atom = Mk(c == '*' ? Re::kStar : c == '+' ? Re::kPlus : Re::kOpt, std::move(atom));
      } else if (c == '{') {
        size_t j = i_ + 1;
        auto number = [&](int* v) {
          if (j >= p_.size() || p_[j] < '0' || p_[j] > '9') return false;
          long n = 0;
          while (j < p_.size() && p_[j] >= '0' && p_[j] <= '9' && n < 100000) n = n * 10 + (p_[j++] - '0');
          *v = static_cast<int>(n);
          return true;
        };
        int lo = 0, hi = -1;
        if (!number(&lo)) return Bad("'{' that is not a repetition count");
        if (j < p_.size() && p_[j] == ',') {
          j++;
          if (j < p_.size() && p_[j] != '}') {
            if (!number(&hi)) return Bad("malformed repetition count");
          }
        } else {
          hi = lo;
        }
        if (j >= p_.size() || p_[j] != '}') return Bad("malformed repetition count");
        if (hi >= 0 && hi < lo) return Bad("repetition count {m,n} with n < m");
        if (lo > 63 || hi > 63 || static_cast<long>(Leaves(*atom)) * std::max(lo, std::max(hi, 1)) > 200)
          return Bad("repetition that needs more than 63 automaton positions");
        i_ = j + 1;
        ReP seq = Mk(Re::kEmpty);
        auto append = [&](ReP x) { seq = seq->kind == Re::kEmpty ? std::move(x) : Mk(Re::kCat, std::move(seq), std::move(x)); };
        for (int k = 0; k < lo; k++) append(Clone(*atom));
        if (hi < 0) append(Mk(Re::kStar, Clone(*atom)));
        for (int k = lo; k < hi; k++) append(Mk(Re::kOpt, Clone(*atom)));
        atom = std::move(seq);
      } else {
        break;
      }
      if (More() && p_[i_] == '?') i_++;  // lazy: the same language
      else if (More() && p_[i_] == '+') return Bad("possessive quantifier");
    }
    *out = std::move(atom);
    return Status::OK();
  }
  // an escape outside or inside a class: one byte set
  Status Escape(ByteSet* set) {
    if (!More()) return Bad("pattern ends in a backslash");
    const char c = p_[i_++];
    switch (c) {
      case 'd': *set = Digits(); return Status::OK();
      case 'w': *set = WordChars(); return Status::OK();
      case 's': *set = PerlSpaces(); return Status::OK();
      case 'D': *set = ~Digits(); negated_escape_ = true; return Status::OK();
      case 'W': *set = ~WordChars(); negated_escape_ = true; return Status::OK();
      case 'S': *set = ~PerlSpaces(); negated_escape_ = true; return Status::OK();
      case 't': *set = Range('\t', '\t'); return Status::OK();
      case 'n': *set = Range('\n', '\n'); return Status::OK();
      case 'r': *set = Range('\r', '\r'); return Status::OK();
      case 'f': *set = Range('\f', '\f'); return Status::OK();
      case 'v': *set = Range('\v', '\v'); return Status::OK();
      case 'x': {
        int v = 0;
        for (int k = 0; k < 2; k++) {
          if (!More()) return Bad("\\x needs two hexadecimal digits");
          const char h = p_[i_++];
          const int d = h >= '0' && h <= '9' ? h - '0' : h >= 'a' && h <= 'f' ? h - 'a' + 10 : h >= 'A' && h <= 'F' ? h - 'A' + 10 : -1;
          if (d < 0) return Bad("\\x needs two hexadecimal digits");
          v = v * 16 + d;
        }
        if (v >= 0x80) return Bad("\\x escape of a byte >= 0x80");
        *set = Range(v, v);
        return Status::OK();
      }
      default:
        if ((c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z') || (c >= '0' && c <= '9')) {
          i_--;
          return Bad(std::string("escape \\") + c);
        }
        *set = Range(static_cast<unsigned char>(c), static_cast<unsigned char>(c));
        return Status::OK();
    }
  }
  Status Class(ReP* out) {  // after '['
    bool negate = false;
    if (More() && p_[i_] == '^') { negate = true; i_++; }
    ByteSet set;
    std::vector<std::string> extras;  // non-ASCII members: whole characters, as alternatives next to the byte set
    bool first = true;
    for (;; first = false) {
      if (!More()) return Bad("unterminated character class");
      unsigned char c = static_cast<unsigned char>(p_[i_]);
      if (c == ']' && !first) { i_++; break; }
      if (c >= 0xC2) {  // a non-ASCII character: a member of its own (not an end of a range)
        size_t j = i_ + 1;
        while (j < p_.size() && (static_cast<unsigned char>(p_[j]) & 0xC0) == 0x80) j++;
        if (j < p_.size() && p_[j] == '-' && j + 1 < p_.size() && p_[j + 1] != ']') return Bad("character range with a non-ASCII end");
        if (fold_) return Bad("(?i) with a non-ASCII character");
        extras.push_back(p_.substr(i_, j - i_));
        i_ = j;
        continue;
      }
      if (c == '[' && i_ + 1 < p_.size() && p_[i_ + 1] == ':') {
        const size_t close = p_.find(":]", i_ + 2);
        if (close == std::string::npos) return Bad("unterminated POSIX character class");
        const std::string name = p_.substr(i_ + 2, close - i_ - 2);
        ByteSet cls;
        if (name == "alpha") cls = Range('a', 'z') | Range('A', 'Z');
        else if (name == "digit") cls = Digits();
        else if (name == "alnum") cls = Digits() | Range('a', 'z') | Range('A', 'Z');
        else if (name == "upper") cls = Range('A', 'Z');
        else if (name == "lower") cls = Range('a', 'z');
        else if (name == "space") cls = Spaces();
        else if (name == "blank") cls = Range(' ', ' ') | Range('\t', '\t');
        else if (name == "punct") cls = Range('!', '/') | Range(':', '@') | Range('[', '`') | Range('{', '~');
        else if (name == "xdigit") cls = Digits() | Range('a', 'f') | Range('A', 'F');
        else if (name == "word") cls = WordChars();
        else if (name == "cntrl") cls = Range(0, 31) | Range(127, 127);
        else if (name == "graph") cls = Range('!', '~');
        else if (name == "print") cls = Range(' ', '~');
        else return Bad("POSIX character class [:" + name + ":]");
        set |= cls;
        i_ = close + 2;
        continue;
      }
      ByteSet one;
      bool single = true;
      i_++;
      if (c == '\\') {
        negated_escape_ = false;
        GDV_RETURN_NOT_OK(Escape(&one));
        if (negated_escape_) return Bad("negated escape inside a character class");
        single = one.count() == 1;
      } else {
        if (c >= 0x80) return Bad("non-ASCII member of a character class");
        one = Range(c, c);
      }
      if (single && i_ + 1 < p_.size() && p_[i_] == '-' && p_[i_ + 1] != ']') {
        int lo = 0;
        while (!one.test(static_cast<size_t>(lo))) lo++;
        i_++;
        unsigned char h = static_cast<unsigned char>(p_[i_++]);
        ByteSet hs;
        if (h == '\\') {
          negated_escape_ = false;
          GDV_RETURN_NOT_OK(Escape(&hs));
          if (hs.count() != 1) return Bad("class escape as the end of a range");
          h = 0;
          while (!hs.test(h)) h++;
        }
        if (h >= 0x80) return Bad("non-ASCII member of a character class");
        if (h < lo) return Bad("character range out of order");
        one = Range(lo, h);
      }
      set |= one;
    }
    if (fold_) set = FoldCase(set);
    if (negate) {
      if (!extras.empty()) return Bad("negated character class with non-ASCII members");
      *out = WholeCharacterOutside(set, fold_);
      return Status::OK();
    }
    ReP all = set.any() || extras.empty() ? (fold_ ? FoldedSet(set) : MkSet(set)) : nullptr;
    for (const std::string& x : extras) {
      ReP seq;
      for (unsigned char k : x) seq = seq ? Mk(Re::kCat, std::move(seq), MkSet(Range(k, k))) : MkSet(Range(k, k));
      all = all ? Mk(Re::kAlt, std::move(all), std::move(seq)) : std::move(seq);
    }
    *out = std::move(all);
    return Status::OK();
  }
  Status Atom(ReP* out) {
    const unsigned char c = static_cast<unsigned char>(p_[i_]);
    if (++atoms_ > kMaxAtoms) return Bad("more than " + std::to_string(kMaxAtoms) + " atoms and quantifiers");
    switch (c) {
      case '(': {
        i_++;
        if (depth_ >= kMaxNesting) return Bad("groups nested deeper than " + std::to_string(kMaxNesting));
        if (More() && p_[i_] == '?') {
          if (i_ + 1 < p_.size() && p_[i_ + 1] == ':') {
            i_ += 2;
          } else if (i_ + 2 < p_.size() && p_[i_ + 1] == 'P' && p_[i_ + 2] == '<') {  // (?P<name>...): a group like any other
            size_t j = i_ + 3;
            while (j < p_.size() && (std::isalnum(static_cast<unsigned char>(p_[j])) || p_[j] == '_')) j++;
            if (j == i_ + 3 || j >= p_.size() || p_[j] != '>') return Bad("malformed group name");
            i_ = j + 1;
          } else {
            return Bad("group flags inside the pattern / look-around");
          }
        }
        depth_++;
        GDV_RETURN_NOT_OK(Alt(out));
        depth_--;
        if (!More() || p_[i_] != ')') return Bad("unmatched '('");
        i_++;
        return Status::OK();
      }
      case '[':
        i_++;
        return Class(out);
      case '.': {
        i_++;
        ByteSet lead = AnyLead();
        if (!dot_nl_) lead.reset('\n');
        *out = WholeCharacter(lead);
        return Status::OK();
      }
      case '\\': {
        i_++;
        if (More() && (p_[i_] == 'b' || p_[i_] == 'B' || p_[i_] == 'A' || p_[i_] == 'z')) {
          const char k = p_[i_++];
          *out = MkAssert(k == 'b' ? 1 : k == 'B' ? 2 : k == 'A' ? 3 : 4);
          return Status::OK();
        }
        ByteSet set;
        negated_escape_ = false;
        GDV_RETURN_NOT_OK(Escape(&set));
        if (fold_) set = negated_escape_ ? ~FoldCase(~set) : FoldCase(set);
        *out = negated_escape_ ? WholeCharacterOutside(~set, fold_) : fold_ ? FoldedSet(set) : MkSet(set);
        return Status::OK();
      }
      case '^':
        i_++;
        *out = MkAssert(3);
        return Status::OK();
      case '$':
        i_++;
        *out = MkAssert(4);
        return Status::OK();
      case '*':
      case '+':
      case '?':
      case '{':
        return Bad("quantifier with nothing to repeat");
      default: {
        i_++;
        if (fold_ && c >= 0x80) return Bad("(?i) with a non-ASCII character");
        ReP atom = fold_ ? FoldedSet(FoldCase(Range(c, c))) : MkSet(Range(c, c));
        // a non-ASCII character of the pattern is ONE atom (a quantifier behind it repeats the character): its lead byte
        // and the continuation bytes that follow, one position each
        if (c >= 0xC2)
          while (More() && (static_cast<unsigned char>(p_[i_]) & 0xC0) == 0x80) {
            const unsigned char k = static_cast<unsigned char>(p_[i_++]);
            atom = Mk(Re::kCat, std::move(atom), MkSet(Range(k, k)));
          }
        *out = std::move(atom);
        return Status::OK();
      }
    }
  }

  const std::string& p_;
  size_t i_ = 0;
  bool negated_escape_ = false;
  int depth_ = 0, atoms_ = 0;
  bool fold_ = false;    // (?i): ASCII letters match in either case
  bool dot_nl_ = false;  // (?s): '.' matches a newline too
};

// Position automaton with GAP CONDITIONS.  An assertion (^ $ \\A \\z \\b \\B) consumes nothing: it constrains the gap between two
// consecutive bytes (or the text's edges).  Every relation of the construction therefore carries a condition: nullable under a
// condition, first / last positions under a condition, follow[p][c] = the positions that may come behind p when the gap between
// them satisfies c.  A condition is a CONJUNCTION of the four predicates — 1 word boundary, 2 not a word boundary, 4 start of
// the text, 8 end of the text — and gets a small id (0 = no predicate); a pattern may use kConds distinct conjunctions
// ("^$" uses three: start, end, start-and-end).
constexpr int kConds = 8;
constexpr int kRegexTableWords = 2 + 3 * kConds + 64 * kConds + 256;  // = GDV_REGEX_TABLE_BYTES / 8
struct Glushkov {
  uint64_t follow[64][kConds] = {};
  ByteSet sets[64];
  int npos = 0;
  struct Info {
    uint32_t nullable = 0;          // bit c: the empty string, where the gap satisfies c
    uint64_t first[kConds] = {};    // first[c]: positions a match may begin with, entered over a gap satisfying c
    uint64_t last[kConds] = {};     // last[c]: positions a match may end with, left over a gap satisfying c
  };
  bool overflow = false, clash = false;
  uint32_t masks[kConds] = {};  // condition id -> its predicates
  int nconds = 1;               // (id 0: no predicate)
  int CondFor(uint32_t mask) {
    for (int c = 0; c < nconds; c++)
      if (masks[c] == mask) return c;
    if (nconds == kConds) { clash = true; return -1; }
    masks[nconds] = mask;
    return nconds++;
  }
  int Both(int a, int b) { return CondFor(masks[a] | masks[b]); }  // the condition "a and b" on one gap
  void Link(const Info& a, const Info& b) {
    for (int ca = 0; ca < kConds; ca++)
      for (int cb = 0; cb < kConds; cb++) {
        if (a.last[ca] == 0 || b.first[cb] == 0) continue;
        const int c = Both(ca, cb);
        if (c < 0) continue;
        for (int p = 0; p < 64; p++)
          if ((a.last[ca] >> p) & 1) follow[p][c] |= b.first[cb];
      }
  }
  Info Build(Re& x) {
    Info r;
    switch (x.kind) {
      case Re::kEmpty:
        r.nullable = 1;
        return r;
      case Re::kAssert: {
        const int c = CondFor(1u << (x.cond - 1));
        if (c >= 0) r.nullable = 1u << c;
        return r;
      }
      case Re::kSet:
        if (npos >= 63) { overflow = true; return r; }
        x.pos = npos++;
        sets[x.pos] = x.set;
        r.first[0] = r.last[0] = 1ull << x.pos;
        return r;
      case Re::kCat: {
        const Info a = Build(*x.a), b = Build(*x.b);
        Link(a, b);
        for (int c = 0; c < kConds; c++) { r.first[c] |= a.first[c]; r.last[c] |= b.last[c]; }
        for (int ca = 0; ca < kConds; ca++)
          for (int cb = 0; cb < kConds; cb++) {
            if (((a.nullable >> ca) & 1) && ((b.nullable >> cb) & 1)) { const int c = Both(ca, cb); if (c >= 0) r.nullable |= 1u << c; }
            if (((a.nullable >> ca) & 1) && b.first[cb] != 0) { const int c = Both(ca, cb); if (c >= 0) r.first[c] |= b.first[cb]; }
            if (((b.nullable >> cb) & 1) && a.last[ca] != 0) { const int c = Both(ca, cb); if (c >= 0) r.last[c] |= a.last[ca]; }
          }
        return r;
      }
      case Re::kAlt: {
        const Info a = Build(*x.a), b = Build(*x.b);
        r.nullable = a.nullable | b.nullable;
        for (int c = 0; c < kConds; c++) { r.first[c] = a.first[c] | b.first[c]; r.last[c] = a.last[c] | b.last[c]; }
        return r;
      }
      case Re::kStar:
      case Re::kPlus:
      case Re::kOpt: {
        r = Build(*x.a);
        if (x.kind != Re::kOpt) Link(r, r);
        if (x.kind != Re::kPlus) r.nullable |= 1;
        return r;
      }
    }
    return r;
  }
};

}  // namespace

Status CompileRegex(const std::string& pattern, std::string* table) {
  // flags, in front of everything only: (?i) ASCII letters in either case, (?s) '.' matches a newline
  std::string body = pattern;
  bool fold = false, dot_nl = false;
  if (body.compare(0, 2, "(?") == 0) {
    size_t j = 2;
    while (j < body.size() && (body[j] == 'i' || body[j] == 's')) j++;
    if (j > 2 && j < body.size() && body[j] == ')') {
      for (size_t k = 2; k < j; k++) (body[k] == 'i' ? fold : dot_nl) = true;
      body.erase(0, j + 1);
    }
  }
  ReP tree;
  Parser parser(body, fold, dot_nl);
  Status st = parser.Parse(&tree);
  if (!st.ok()) {  // (messages quote the pattern as the caller wrote it)
    const std::string quoted = "'" + body + "'";
    const size_t at = st.msg.find(quoted);
    if (at != std::string::npos && body != pattern) st.msg.replace(at, quoted.size(), "'" + pattern + "'");
    return st;
  }
  Glushkov g;
  const Glushkov::Info top = g.Build(*tree);
  const std::string prefix = "regular expression '" + pattern + "' not supported yet by the HIP backend: ";
  if (g.overflow) return Status::CodeGenError(prefix + "more than 63 automaton positions");
  if (g.clash) return Status::CodeGenError(prefix + "more than 7 distinct combinations of assertions (^ $ \\A \\z \\b \\B)");
  // layout (64-bit words): flags | nullable | predicates[8] | first[8] | last[8] | follow[64][8] | match[256]
  //   flags bit 0: every way into the pattern asks for the start of the text (nothing can begin behind byte 0);
  //   bits 8..15: the condition ids that occur at all (the row evaluates only those)
  //   predicates[c]: the conjunction condition c stands for — 1 word boundary, 2 not one, 4 start of the text, 8 end of it
  table->assign(kRegexTableWords * 8, '\0');
  uint64_t* tb = reinterpret_cast<uint64_t*>(&(*table)[0]);
  uint64_t* const first = tb + 2 + kConds;
  uint64_t* const last = first + kConds;
  uint64_t* const follow = last + kConds;
  uint64_t* const match = follow + 64 * kConds;
  uint64_t used = 1;
  bool start_only = true;
  for (int c = 0; c < kConds; c++) {
    tb[2 + c] = g.masks[c];
    first[c] = top.first[c];
    last[c] = top.last[c];
    if (top.first[c] != 0 || top.last[c] != 0 || ((top.nullable >> c) & 1)) used |= 1ull << c;
    if ((g.masks[c] & 4u) == 0 && (top.first[c] != 0 || ((top.nullable >> c) & 1))) start_only = false;
  }
  for (int p = 0; p < 64; p++)
    for (int c = 0; c < kConds; c++) {
      follow[p * kConds + c] = g.follow[p][c];
      if (g.follow[p][c] != 0) used |= 1ull << c;
    }
  tb[0] = (start_only ? 1u : 0u) | (used << 8);
  tb[1] = top.nullable;
  for (int b = 0; b < 256; b++) {
    uint64_t m = 0;
    for (int p = 0; p < g.npos; p++)
      if (g.sets[p].test(static_cast<size_t>(b))) m |= 1ull << p;
    match[b] = m;
  }
  return Status::OK();
}

// to_date's SQL pattern -> one byte per strptime directive (gdv_parse_date).  Tokens are matched case-insensitively,
// longest first; any other letter sequence is an error, every other character stands for itself (white space: any
// run of it) [date_utils.cc DateUtils::ToInternalFormat, as recalled; the time-zone tokens TZD / TZO / TZH:TZM and the
// fractional-second / era / century / week-of-year tokens are not taken: the message says so].
Status CompileDateFormat(const std::string& pattern, std::string* ops) {
  static const std::pair<const char*, char> tokens[] = {
      {"YYYY", 'Y'}, {"HH24", 'H'}, {"HH12", 'I'}, {"MONTH", 'b'}, {"MON", 'b'}, {"DDD", 'j'}, {"DAY", 'a'}, {"YY", 'y'}, {"MM", 'm'},
      {"DD", 'd'},   {"DY", 'a'},   {"HH", 'I'},   {"MI", 'M'},    {"SS", 'S'},  {"AM", 'p'},  {"PM", 'p'}};
  ops->clear();
  bool quoted = false;  // inside "double quotes" every character stands for itself
  for (size_t i = 0; i < pattern.size();) {
    const unsigned char c = static_cast<unsigned char>(pattern[i]);
    if (c == '"') { quoted = !quoted; i++; continue; }
    if (quoted) { ops->push_back('L'); ops->push_back(static_cast<char>(c)); i++; continue; }
    if (c == ' ' || (c >= 9 && c <= 13)) {
      if (ops->empty() || ops->back() != ' ') ops->push_back(' ');
      i++;
      continue;
    }
    if (!((c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z'))) {
      ops->push_back('L');
      ops->push_back(static_cast<char>(c));
      i++;
      continue;
    }
    bool hit = false;
    for (auto& tk : tokens) {
      const size_t n = std::strlen(tk.first);
      if (i + n > pattern.size()) continue;
      bool same = true;
      for (size_t j = 0; same && j < n; j++) same = (pattern[i + j] & ~0x20) == tk.first[j] || pattern[i + j] == tk.first[j];
      if (same) { ops->push_back(tk.second); i += n; hit = true; break; }
    }
    if (!hit) {
      size_t j = i;
      while (j < pattern.size() && (((pattern[j] | 0x20) >= 'a' && (pattern[j] | 0x20) <= 'z') || (pattern[j] >= '0' && pattern[j] <= '9'))) j++;
      return Status::Invalid("Invalid date format: the HIP backend takes YYYY YY MM MON MONTH DD DDD DY DAY HH HH12 HH24 MI SS AM PM; not '" +
                             pattern.substr(i, j - i) + "'");
    }
  }
  return Status::OK();
}

}  // namespace gdv
