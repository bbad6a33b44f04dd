// Extension class the DuckDB loader instantiates; same name and shape as the reference's
// (duckdb_extension/src/include/duckdb_imputation_extension.hpp:7-11).
#pragma once

#include "duckdb.hpp"

namespace duckdb {

class DuckdbImputationExtension : public Extension {
public:
  void Load(DuckDB &db) override;
  std::string Name() override;
};

}  // namespace duckdb
