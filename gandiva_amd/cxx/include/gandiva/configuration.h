// gandiva/configuration.h (pyarrow/includes/libgandiva.pxd:279-298).  `optimize` and
// `dump_ir` are kept for source compatibility; the HIP backend always builds -O3 and
// DumpIR() is always available.
#pragma once
#include <memory>

namespace gandiva {
class Configuration {
 public:
  Configuration() = default;
  Configuration(bool optimize, bool dump_ir) : optimize_(optimize), dump_ir_(dump_ir) {}
  bool optimize() const { return optimize_; }
  bool dump_ir() const { return dump_ir_; }
  void set_optimize(bool v) { optimize_ = v; }
  void set_dump_ir(bool v) { dump_ir_ = v; }

 private:
  bool optimize_ = true;
  bool dump_ir_ = false;
};

class ConfigurationBuilder {
 public:
  std::shared_ptr<Configuration> build() { return std::make_shared<Configuration>(); }
  static std::shared_ptr<Configuration> DefaultConfiguration() {
    static std::shared_ptr<Configuration> c = std::make_shared<Configuration>();
    return c;
  }
};
}  // namespace gandiva
