// Regular expressions for regexp_like / regexp_matches (round 5): the pattern — a literal of the expression, as for the
// reference's holders, which compile it with RE2 once per expression [regexp_matches_holder.cc / like_holder.cc, as
// recalled] — is compiled HERE, at Make time, into a position automaton (Glushkov: one state per character position of the
// pattern, no epsilon moves) of at most 63 positions, laid out as a table the device function gdv_regex_search walks with
// one 64-bit state set per row.  Assertions consume nothing: they become CONDITIONS on the automaton's edges — conjunctions of
// word boundary / not a word boundary / start of the text / end of the text, at most 8 distinct ones per pattern — evaluated on
// the gap between two bytes:
//     uint64 flags, nullable, predicates[8], first[8], last[8], follow[64][8], match[256]     (GDV_REGEX_TABLE_BYTES = 6352)
// (match[b] = the positions that accept byte b; flags bit 0: every way in asks for the start of the text, bits 8-15: the
// conditions that occur; predicates[c]: what condition c asks).  "Does the text contain a match" is all the two functions ask, so greedy / lazy / leftmost-first
// make no difference and the automaton's answer is RE2's (tests hold it to the RE2 in this image's libarrow).
//
// Syntax taken: literals, '.', classes [a-z0-9_] and [^...] (ASCII members; non-ASCII characters as members of a positive
// class), POSIX classes [[:alpha:]] ... inside brackets, \d \D \w \W \s \S, \t \n \r \f \v \xHH and escaped punctuation,
// groups ( ), (?: ) and (?P<name> ), alternation |, quantifiers * + ? {m} {m,} {m,n} (and their lazy forms), the assertions
// ^ $ \A \z \b \B anywhere, the flags (?i) (ASCII letters in either case — and, as RE2's simple folding has it, U+212A KELVIN
// SIGN with k and U+017F LONG S with s, in positive and negated sets alike) and (?s) ('.' matches a newline) in front of
// everything.  \s / \S are Perl's [\t\n\f\r ] (no vertical tab); [[:space:]] holds it.  '.' and negated classes consume a whole UTF-8 character (a lead byte and its continuation bytes); \b is
// RE2's ASCII word boundary.  Anything else — back-references and look-around (RE2 has neither), flags inside the pattern,
// Unicode classes \p{..}, non-ASCII range ends or negated classes with non-ASCII members, (?i) next to non-ASCII characters,
// more than 7 distinct combinations of assertions, more than 63 positions — is refused with a message, not guessed; so are
// patterns beyond 4096 bytes, 64 levels of nesting or 512 atoms and quantifiers (the tree is walked recursively; patterns come from user SQL).
// Texts that are not valid UTF-8: '.' and negated classes take "a lead byte and whatever continuation bytes follow" as one
// character, which RE2 does not — the two agree on valid UTF-8 only.
#pragma once
#include <string>

#include "gdv_types.h"

namespace gdv {

// table = the bytes described above; CodeGenError with the reason otherwise
Status CompileRegex(const std::string& pattern, std::string* table);

// to_date's SQL pattern -> one byte per strptime directive, the program gdv_parse_date interprets (gdv_regex.cc)
Status CompileDateFormat(const std::string& pattern, std::string* ops);

}  // namespace gdv
