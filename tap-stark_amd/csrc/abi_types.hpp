// The opaque handle types of include/tapstark.h, shared by the translation units that implement
// the C ABI (abi.cpp, taptree.cpp, comm.cpp).
#pragma once
#include <memory>

#include "../../include/tapstark.h"
#include "host.hpp"

struct ts_ctx {
    ts::Context ctx;
    explicit ts_ctx(int dev) : ctx(dev) {}
};
struct ts_matrix {
    ts::DeviceMatrix m;
};
struct ts_pcs_data {
    std::unique_ptr<ts::PcsData> d;
};
