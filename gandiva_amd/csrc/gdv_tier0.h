// Tier 0 (round 6): a post-fix program for the ahead-of-time interpreter kernel (gdv_tier0.hip) — what a Projector /
// Filter evaluates with while hipRTC compiles its specialised kernel.  Built at Make from the same expression trees, over
// the same argument block as the generated kernel; covers the fixed-width core of the registry (add / subtract / multiply,
// the six comparisons, not / isnull / isnotnull, the numeric casts, if / else, AND / OR, literals) — plans outside it
// have no tier 0 and wait for their compilation as before.
#pragma once
#include <cstdint>

namespace gdv {
namespace tier0 {

enum TypeKind : int { kTBool = 0, kTI8, kTU8, kTI16, kTU16, kTI32, kTU32, kTI64, kTU64, kTF32, kTF64 };
enum Op : int {
  kLoad = 1, kLit, kAdd, kSub, kMul, kCmp, kCast, kNot, kIsNull, kIsNotNull, kAnd2, kOr2, kIf, kOut, kFilterOut
};
enum Cmp : int { kEq = 0, kNe, kLt, kLe, kGt, kGe };

constexpr int kMaxDepth = 12;     // operand stack entries (LDS: 4 waves x 12 x 64 x 8 bytes = 24 KiB per workgroup)
constexpr int kMaxCode = 256;     // instructions: op | a << 8 | b << 16 | c << 24
constexpr int kMaxLits = 32;
constexpr int kMaxBlock = 2048;   // bytes of the argument block (ArgLayout::total())

// passed BY VALUE (kernel arguments: 3.3 KiB of the 4 KiB a launch may carry) — no allocation, no upload
struct Args {
  uint8_t block[kMaxBlock] __attribute__((aligned(8)));  // the generated kernel's own argument block (ArgLayout)
  uint32_t code[kMaxCode];
  uint64_t lits[kMaxLits];
  int32_t ncode, n_in, filter, subtiles;
};

}  // namespace tier0
}  // namespace gdv

// ---- host side (gdv_tier0.cc): the program of a plan, or "this plan has no tier 0" ----------------------------------
#ifndef __HIP_DEVICE_COMPILE__
#include <hip/hip_runtime_api.h>

#include <string>
#include <vector>

#include "gdv_node.h"
#include "gdv_planner.h"

namespace gdv {

// Builds the program for `exprs` (projector outputs in order) or for a filter's condition (filter = true: exprs holds the
// one condition) over the argument block layout and input slots of `plan`.  False — with the reason in *why — when a
// node, a type or the plan's shape is outside what the interpreter takes; the caller then has no tier 0 for this plan.
bool BuildTier0Program(const Schema& schema, const std::vector<ExpressionPtr>& exprs, bool filter, const KernelPlan& plan,
                       tier0::Args* out, std::string* why);

// the program as text, one instruction per line (diagnostics, tests)
std::string DescribeTier0Program(const tier0::Args& prog);
// evaluations that ran on tier 0 (process-wide)
int64_t Tier0Launches();
void CountTier0Launch();

// gdv_tier0.hip
hipError_t LaunchTier0(const tier0::Args& args, int64_t rows, int num_cus, hipStream_t stream);

}  // namespace gdv
#endif
