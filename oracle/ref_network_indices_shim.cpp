// TEST INFRASTRUCTURE ONLY: the host code that lays out the network buffer of a model-parallel
// embedding_collection group -- NetworkIndices::init, R/HugeCTR/embedding/operators/
// network_forward.cu:23-62 (up to where it copies its four vectors to the device) -- cut out of the
// checkout by oracle/Makefile and compiled as it stands: for every destination lookup the list of
// (source GPU, index of the lookup among that GPU's local lookups) whose pooled partial vectors
// NetworkForward adds up.  The struct is declared here with the four host vectors it fills
// (network_forward.hpp:24-37 without the device tensors).
#include <algorithm>
#include <memory>
#include <tuple>
#include <vector>

class CoreResourceManager;
namespace embedding {
struct NetworkIndices {
  std::vector<int> h_network_ids;
  std::vector<int> h_network_gpu_ids;
  std::vector<int> h_network_offsets;
  std::vector<int> h_network_dst_lookup_ids;
  void init(std::shared_ptr<CoreResourceManager> core,
            const std::vector<std::vector<int>>& h_global_lookup_ids);
};
#include "_ref/gen/network_indices_init.inc"
}  // namespace embedding

extern "C" {
// lookups [offsets[g], offsets[g + 1]) = the local lookup ids of GPU g.  Outputs sized by the caller:
// ids / gpu_ids [total], net_offsets [n_dst + 1], dst_lookup_ids [n_dst]; returns n_dst
int refnet_indices(int num_gpus, const int* offsets, const int* lookups, int* ids, int* gpu_ids,
                   int* net_offsets, int* dst_lookup_ids) {
  std::vector<std::vector<int>> h(num_gpus);
  for (int g = 0; g < num_gpus; g++) h[g].assign(lookups + offsets[g], lookups + offsets[g + 1]);
  embedding::NetworkIndices n;
  n.init(nullptr, h);
  std::copy(n.h_network_ids.begin(), n.h_network_ids.end(), ids);
  std::copy(n.h_network_gpu_ids.begin(), n.h_network_gpu_ids.end(), gpu_ids);
  std::copy(n.h_network_offsets.begin(), n.h_network_offsets.end(), net_offsets);
  std::copy(n.h_network_dst_lookup_ids.begin(), n.h_network_dst_lookup_ids.end(), dst_lookup_ids);
  return (int)n.h_network_dst_lookup_ids.size();
}
}
