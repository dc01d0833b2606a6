// kafka.hpp — host side of SURVEY.md §8 f-4 (second half): Kafka payloads -> messages.
//
// What the reference's decodeKafkaPayload (aggregator/data.go:929-1017) yields for one L7 event: the records of
// every RecordBatch of a Produce request (api key 0, request.go:28-62, produce_request.go:29-89) or of a Fetch
// response (response_header.go:295-313, fetch_response.go:41-214).  One packed event is emitted per message
// (processKafkaEvent :1035-1079); an event whose decode fails — or panics, which the reference recovers from
// (:943-948) — or yields no message is dropped.  Record batches are CRC-32C checked and may be gzip / snappy
// (xerial framed or raw) / lz4-frame / zstd compressed (record_batch.go:51-139, decompress.go:40-98); legacy
// message sets are walked too, but a payload that still carries one ends in the reference's nil-RecordBatch
// panic, i.e. in no message.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace alaz {
namespace kafka {

struct Message { std::string Topic; int32_t Partition = 0; std::string Key, Value; };

enum class Status { kOk = 0, kInsufficientData = 1, kError = 2, kPanic = 3 };

// method_id: 1 = PRODUCE_REQUEST, 2 = FETCH_RESPONSE (ebpf/c/kafka.c:15-16); api_version is the request's
// (carried to the response event by the kernel side, l7.c:869).  Messages are ordered by (topic, partition,
// position); the reference iterates Go maps, i.e. in no defined order.
// out == nullptr: only *count is produced (what the packer needs: one packed event per message) and no key / value
// is copied.
Status DecodePayload(const uint8_t* payload, size_t size, int method_id, int16_t api_version, std::vector<Message>* out, size_t* count = nullptr);

// codec 0 none, 1 gzip, 2 snappy, 3 lz4, 4 zstd.  *nil_slice: the Go decoder would have returned a nil slice.
bool Decompress(int codec, const uint8_t* src, size_t n, std::string* out, bool* nil_slice = nullptr);

uint32_t Crc32(const uint8_t* p, size_t n, bool castagnoli);
uint32_t XXH32(const uint8_t* p, size_t n, uint32_t seed);

}  // namespace kafka
}  // namespace alaz
