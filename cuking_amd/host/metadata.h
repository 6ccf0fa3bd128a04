// metadata.json reader: {"num_sites": <int>, "samples": [<string>, ...]}
// (written by mt_to_cuking_inputs.py:40-47, read at cuking.cu:475-500).
// A small strict JSON parser; no third-party JSON library is available.
#ifndef CUKING_AMD_HOST_METADATA_H_
#define CUKING_AMD_HOST_METADATA_H_

#include <cstdint>
#include <string>
#include <vector>

namespace cuking_host {

struct Metadata {
  uint32_t num_sites = 0;
  std::vector<std::string> samples;
};

// Returns "" on success, otherwise the error message.
std::string ParseMetadata(const std::string &json_text, Metadata *out);
std::string ReadMetadataFile(const std::string &path, Metadata *out);

}  // namespace cuking_host

#endif  // CUKING_AMD_HOST_METADATA_H_
