// communicator of the multi-GPU commitments (comm.cpp); owned by the context
#pragma once
struct plk_ctx;
namespace plk {
void comm_release(plk_ctx *ctx);
}
