#include "gdv_regex.h"

#include <algorithm>
#include <array>
#include <bitset>
#include <cstring>
#include <memory>
#include <vector>

namespace gdv {
namespace {

using ByteSet = std::bitset<256>;

// syntax tree: leaves are byte sets (one automaton position each)
struct Re {
  enum Kind { kEmpty, kSet, kCat, kAlt, kStar, kPlus, kOpt } kind = kEmpty;
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
ReP Clone(const Re& x) {
  ReP r(new Re);
  r->kind = x.kind;
  r->set = x.set;
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
ByteSet Spaces() {
  ByteSet s;
  for (char c : {' ', '\t', '\n', '\r', '\f', '\v'}) s.set(static_cast<unsigned char>(c));
  return s;
}

// (?i): a set that holds an ASCII letter holds it in both cases
ByteSet FoldCase(ByteSet s) {
  for (int c = 'a'; c <= 'z'; c++)
    if (s.test(static_cast<size_t>(c)) || s.test(static_cast<size_t>(c - 32))) { s.set(static_cast<size_t>(c)); s.set(static_cast<size_t>(c - 32)); }
  return s;
}

class Parser {
 public:
  Parser(const std::string& p, bool fold) : p_(p), fold_(fold) {}
  Status Parse(ReP* out) {
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
  Status Alt(ReP* out) {
    ReP left;
    GDV_RETURN_NOT_OK(Cat(&left));
    while (More() && p_[i_] == '|') {
      i_++;
      ReP right;
      GDV_RETURN_NOT_OK(Cat(&right));
      left = Mk(Re::kAlt, std::move(left), std::move(right));
    }
    *out = std::move(left);
    return Status::OK();
  }
  Status Cat(ReP* out) {
    ReP left = Mk(Re::kEmpty);
    while (More() && p_[i_] != '|' && p_[i_] != ')') {
      ReP piece;
      GDV_RETURN_NOT_OK(Repeat(&piece));
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
      case 's': *set = Spaces(); return Status::OK();
      case 'D': *set = ~Digits(); negated_escape_ = true; return Status::OK();
      case 'W': *set = ~WordChars(); negated_escape_ = true; return Status::OK();
      case 'S': *set = ~Spaces(); negated_escape_ = true; return Status::OK();
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
      if (c == '[' && i_ + 1 < p_.size() && p_[i_ + 1] == ':') return Bad("POSIX character class");
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
      *out = WholeCharacter(AnyLead() & ~set);
      return Status::OK();
    }
    ReP all = set.any() || extras.empty() ? MkSet(set) : nullptr;
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
    switch (c) {
      case '(': {
        i_++;
        if (More() && p_[i_] == '?') {
          if (i_ + 1 < p_.size() && p_[i_ + 1] == ':') i_ += 2;
          else return Bad("group flags / look-around / named groups");
        }
        GDV_RETURN_NOT_OK(Alt(out));
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
        lead.reset('\n');
        *out = WholeCharacter(lead);
        return Status::OK();
      }
      case '\\': {
        i_++;
        ByteSet set;
        negated_escape_ = false;
        GDV_RETURN_NOT_OK(Escape(&set));
        if (fold_) set = negated_escape_ ? ~FoldCase(~set) : FoldCase(set);
        *out = negated_escape_ ? WholeCharacter(AnyLead() & set) : MkSet(set);
        return Status::OK();
      }
      case '^':
      case '$':
        return Bad("an anchor that is not the pattern's first ('^') or last ('$') character");
      case '*':
      case '+':
      case '?':
      case '{':
        return Bad("quantifier with nothing to repeat");
      default: {
        i_++;
        if (fold_ && c >= 0x80) return Bad("(?i) with a non-ASCII character");
        ReP atom = MkSet(fold_ ? FoldCase(Range(c, c)) : Range(c, c));
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
  bool fold_ = false;  // (?i): ASCII letters match in either case
};

struct Glushkov {
  uint64_t follow[64] = {};
  ByteSet sets[64];
  int npos = 0;
  struct Info {
    bool nullable;
    uint64_t first, last;
  };
  bool overflow = false;
  Info Build(Re& x) {
    switch (x.kind) {
      case Re::kEmpty: return {true, 0, 0};
      case Re::kSet: {
        if (npos >= 63) { overflow = true; return {false, 0, 0}; }
        x.pos = npos++;
        sets[x.pos] = x.set;
        return {false, 1ull << x.pos, 1ull << x.pos};
      }
      case Re::kCat: {
        const Info a = Build(*x.a), b = Build(*x.b);
        Link(a.last, b.first);
        return {a.nullable && b.nullable, a.first | (a.nullable ? b.first : 0), b.last | (b.nullable ? a.last : 0)};
      }
      case Re::kAlt: {
        const Info a = Build(*x.a), b = Build(*x.b);
        return {a.nullable || b.nullable, a.first | b.first, a.last | b.last};
      }
      case Re::kStar:
      case Re::kPlus: {
        const Info a = Build(*x.a);
        Link(a.last, a.first);
        return {x.kind == Re::kStar || a.nullable, a.first, a.last};
      }
      case Re::kOpt: {
        const Info a = Build(*x.a);
        return {true, a.first, a.last};
      }
    }
    return {true, 0, 0};
  }
  void Link(uint64_t from, uint64_t to) {
    for (int p = 0; p < 64; p++)
      if ((from >> p) & 1) follow[p] |= to;
  }
};

}  // namespace

Status CompileRegex(const std::string& pattern, std::string* table) {
  // the two anchors the backend takes: '^' first, '$' last (not escaped)
  std::string body = pattern;
  uint64_t flags = 0;
  bool fold = false;
  if (body.compare(0, 4, "(?i)") == 0) {  // the one flag taken, and only in front: ASCII letters in either case
    fold = true;
    body.erase(0, 4);
  }
  if (!body.empty() && body.front() == '^') {
    flags |= 2;
    body.erase(0, 1);
  }
  if (!body.empty() && body.back() == '$') {
    size_t slashes = 0;
    while (slashes + 1 < body.size() && body[body.size() - 2 - slashes] == '\\') slashes++;
    if (slashes % 2 == 0) {
      flags |= 4;
      body.pop_back();
    }
  }
  if (flags != 0) {
    // '^a|b' anchors its first branch only: not expressible with one flag for the whole pattern
    int depth = 0;
    bool in_class = false;
    for (size_t i = 0; i < body.size(); i++) {
      const char c = body[i];
      if (c == '\\') { i++; continue; }
      if (in_class) { if (c == ']') in_class = false; continue; }
      if (c == '[') { in_class = true; if (i + 1 < body.size() && body[i + 1] == '^') i++; if (i + 1 < body.size() && body[i + 1] == ']') i++; }
      else if (c == '(') depth++;
      else if (c == ')') depth--;
      else if (c == '|' && depth == 0)
        return Status::CodeGenError("regular expression '" + pattern + "' not supported yet by the HIP backend: an anchor next to a top-level '|' "
                                    "(write ^(a|b)$)");
    }
  }
  ReP tree;
  Parser parser(body, fold);
  Status st = parser.Parse(&tree);
  if (!st.ok()) {  // (messages quote the pattern as the caller wrote it)
    const std::string quoted = "'" + body + "'";
    const size_t at = st.msg.find(quoted);
    if (at != std::string::npos && body != pattern) st.msg.replace(at, quoted.size(), "'" + pattern + "'");
    return st;
  }
  Glushkov g;
  const Glushkov::Info top = g.Build(*tree);
  if (g.overflow)
    return Status::CodeGenError("regular expression '" + pattern + "' not supported yet by the HIP backend: more than 63 automaton positions");
  if (top.nullable) flags |= 1;
  table->assign((3 + 64 + 256) * 8, '\0');
  uint64_t* t = reinterpret_cast<uint64_t*>(&(*table)[0]);
  t[0] = top.first;
  t[1] = top.last;
  t[2] = flags;
  for (int p = 0; p < 64; p++) t[3 + p] = g.follow[p];
  for (int b = 0; b < 256; b++) {
    uint64_t m = 0;
    for (int p = 0; p < g.npos; p++)
      if (g.sets[p].test(static_cast<size_t>(b))) m |= 1ull << p;
    t[3 + 64 + b] = m;
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
