// mapping::proto::HybridGrid wire format <-> device grid
// (mapping/proto/3d/hybrid_grid.proto; HybridGrid::ToProto hybrid_grid.h:530-542, the proto
// constructor hybrid_grid.h:475-486).  proto3 message, serialised in field order with packed
// repeated scalars -- what the C++ and Python protobuf runtimes emit for it:
//   1: float resolution (fixed32)            3,4,5: packed sint32 (zigzag varint) x/y/z indices
//   6: packed int32 (varint) values
// Cells appear in the reference iterator's order (hybrid_grid.h:93-127,187-231,303-371): 64^3 meta
// cells z-major, 8^3 leaves z-major inside a meta cell, 512 cells z-major inside a leaf, zero
// cells skipped -- independent of DynamicGrid::bits_ because the index shift is a multiple of 64.
#include <algorithm>
#include <array>
#include <cmath>
#include <cstring>
#include <map>
#include <vector>

#include "../../include/dliom.h"

namespace {

void put_varint(std::vector<uint8_t>* out, uint64_t v) {
  while (v >= 0x80) {
    out->push_back(static_cast<uint8_t>(v) | 0x80);
    v >>= 7;
  }
  out->push_back(static_cast<uint8_t>(v));
}
inline uint32_t zigzag(int32_t n) { return (static_cast<uint32_t>(n) << 1) ^ static_cast<uint32_t>(n >> 31); }
inline int32_t unzigzag(uint32_t n) { return static_cast<int32_t>((n >> 1) ^ (~(n & 1) + 1)); }

void put_packed(std::vector<uint8_t>* out, int field, const std::vector<uint8_t>& payload) {
  if (payload.empty()) return;  // proto3: empty repeated fields are not emitted
  put_varint(out, static_cast<uint64_t>(field) << 3 | 2);
  put_varint(out, payload.size());
  out->insert(out->end(), payload.begin(), payload.end());
}

bool get_varint(const uint8_t*& p, const uint8_t* end, uint64_t* v) {
  *v = 0;
  for (int shift = 0; shift < 64 && p < end; shift += 7) {
    const uint8_t b = *p++;
    *v |= static_cast<uint64_t>(b & 0x7F) << shift;
    if ((b & 0x80) == 0) return true;
  }
  return false;
}

// probability_values.{h,cc}: the round trip the proto constructor applies to every value
inline float clampf(float v, float lo, float hi) { return v < lo ? lo : (v > hi ? hi : v); }
uint16_t set_probability_of_value(int32_t proto_value) {
  const float kMin = 0.1f, kMax = 1.f - 0.1f;
  const uint16_t v16 = static_cast<uint16_t>(proto_value);  // ValueToProbability(uint16)
  const int v = v16 & 0x7FFF;                               // the table repeats above the update marker
  float p = kMin;
  if (v != 0) {
    const float kScale = (kMax - kMin) / 32766.f;
    p = v * kScale + (kMin - kScale);
  }
  return static_cast<uint16_t>(static_cast<int>(std::lround((clampf(p, kMin, kMax) - kMin) * (32766.f / (kMax - kMin)))) + 1);
}

}  // namespace

extern "C" int dliom_grid_to_proto(const dliom_grid* grid, uint8_t* buffer, int64_t capacity, int64_t* size) {
  if (grid == nullptr || size == nullptr || capacity < 0) return DLIOM_ERR_INVALID_ARGUMENT;
  int64_t n = 0;
  int s = dliom_grid_num_blocks(grid, &n);
  if (s != DLIOM_OK) return s;
  std::vector<int32_t> origins(static_cast<size_t>(3 * n));
  std::vector<uint16_t> values(static_cast<size_t>(512 * n));
  if (n > 0) {
    s = dliom_grid_download_blocks(grid, origins.data(), values.data(), n, &n);
    if (s != DLIOM_OK) return s;
  }
  // the iterator's order: (meta z, y, x), then (leaf z, y, x): floor-divide the leaf origin
  std::vector<int64_t> order(static_cast<size_t>(n));
  for (int64_t i = 0; i < n; ++i) order[i] = i;
  auto key = [&](int64_t i) {
    const int32_t* o = &origins[3 * i];
    auto fdiv = [](int a, int b) { return (a >= 0 ? a : a - b + 1) / b; };
    const int mx = fdiv(o[0], 64), my = fdiv(o[1], 64), mz = fdiv(o[2], 64);
    const int lx = (o[0] - 64 * mx) >> 3, ly = (o[1] - 64 * my) >> 3, lz = (o[2] - 64 * mz) >> 3;
    // |cell| <= 8192 (bits <= 8) -> |meta| <= 128: 10 biased bits per meta coordinate, 3 per leaf coordinate
    return (static_cast<uint64_t>(mz + 512) << 29) | (static_cast<uint64_t>(my + 512) << 19) |
           (static_cast<uint64_t>(mx + 512) << 9) | (static_cast<uint64_t>(lz) << 6) |
           (static_cast<uint64_t>(ly) << 3) | static_cast<uint64_t>(lx);
  };
  std::sort(order.begin(), order.end(), [&](int64_t a, int64_t b) { return key(a) < key(b); });
  std::vector<uint8_t> xs, ys, zs, vs;
  for (int64_t k = 0; k < n; ++k) {
    const int64_t i = order[k];
    const int32_t* o = &origins[3 * i];
    const uint16_t* leaf = &values[512 * i];
    for (int c = 0; c < 512; ++c) {
      const uint16_t v = leaf[c];
      if (v == 0) continue;
      if (v >= 32768) return DLIOM_ERR_INVALID_ARGUMENT;  // CHECK(update_indices_.empty())
      put_varint(&xs, zigzag(o[0] + (c & 7)));
      put_varint(&ys, zigzag(o[1] + ((c >> 3) & 7)));
      put_varint(&zs, zigzag(o[2] + (c >> 6)));
      put_varint(&vs, v);
    }
  }
  std::vector<uint8_t> out;
  float resolution = 0.f;
  s = dliom_grid_resolution(grid, &resolution);
  if (s != DLIOM_OK) return s;
  uint32_t bits;
  std::memcpy(&bits, &resolution, 4);
  if (bits != 0) {  // proto3 default elision
    out.push_back(0x0D);
    for (int b = 0; b < 4; ++b) out.push_back(static_cast<uint8_t>(bits >> (8 * b)));
  }
  put_packed(&out, 3, xs);
  put_packed(&out, 4, ys);
  put_packed(&out, 5, zs);
  put_packed(&out, 6, vs);
  *size = static_cast<int64_t>(out.size());
  if (buffer == nullptr) return DLIOM_OK;  // size query
  if (capacity < *size) return DLIOM_ERR_CAPACITY;
  std::memcpy(buffer, out.data(), out.size());
  return DLIOM_OK;
}

extern "C" int dliom_grid_from_proto(dliom_ctx* ctx, const uint8_t* buffer, int64_t size, dliom_grid** out) {
  if (ctx == nullptr || out == nullptr || size < 0 || (size > 0 && buffer == nullptr)) return DLIOM_ERR_INVALID_ARGUMENT;
  *out = nullptr;
  float resolution = 0.f;
  std::vector<int32_t> idx[3];
  std::vector<int32_t> vals;
  const uint8_t* p = buffer;
  const uint8_t* end = buffer + size;
  while (p < end) {
    uint64_t tag;
    if (!get_varint(p, end, &tag)) return DLIOM_ERR_INVALID_ARGUMENT;
    const int field = static_cast<int>(tag >> 3), wire = static_cast<int>(tag & 7);
    auto take = [&](uint64_t raw) {
      if (field >= 3 && field <= 5) idx[field - 3].push_back(unzigzag(static_cast<uint32_t>(raw)));
      if (field == 6) vals.push_back(static_cast<int32_t>(raw));
    };
    if (wire == 0) {  // unpacked element of a repeated field (parsers must accept both encodings)
      uint64_t v;
      if (!get_varint(p, end, &v)) return DLIOM_ERR_INVALID_ARGUMENT;
      take(v);
    } else if (wire == 5) {
      if (end - p < 4) return DLIOM_ERR_INVALID_ARGUMENT;
      if (field == 1) std::memcpy(&resolution, p, 4);
      p += 4;
    } else if (wire == 1) {
      if (end - p < 8) return DLIOM_ERR_INVALID_ARGUMENT;
      p += 8;
    } else if (wire == 2) {
      uint64_t len;
      if (!get_varint(p, end, &len) || static_cast<uint64_t>(end - p) < len) return DLIOM_ERR_INVALID_ARGUMENT;
      const uint8_t* q = p;
      const uint8_t* qe = p + len;
      if (field >= 3 && field <= 6) {
        while (q < qe) {
          uint64_t v;
          if (!get_varint(q, qe, &v)) return DLIOM_ERR_INVALID_ARGUMENT;
          take(v);
        }
      }
      p = qe;
    } else {
      return DLIOM_ERR_INVALID_ARGUMENT;  // groups are not part of this message
    }
  }
  const size_t n = vals.size();
  if (idx[0].size() != n || idx[1].size() != n || idx[2].size() != n) return DLIOM_ERR_INVALID_ARGUMENT;  // CHECK_EQ x3
  if (!(resolution > 0.f)) return DLIOM_ERR_INVALID_ARGUMENT;
  int s = dliom_grid_create(ctx, resolution, out);
  if (s != DLIOM_OK) return s;
  if (n > 0) {
    // sequential SetProbability: the last entry for a cell wins
    std::map<std::array<int32_t, 3>, uint16_t> cells;
    for (size_t i = 0; i < n; ++i) cells[{{idx[0][i], idx[1][i], idx[2][i]}}] = set_probability_of_value(vals[i]);
    std::vector<int32_t> xyz;
    std::vector<uint16_t> v;
    xyz.reserve(3 * cells.size());
    v.reserve(cells.size());
    for (const auto& kv : cells) {
      xyz.insert(xyz.end(), kv.first.begin(), kv.first.end());
      v.push_back(kv.second);
    }
    s = dliom_grid_set_values(*out, xyz.data(), v.data(), static_cast<int64_t>(v.size()));
    if (s != DLIOM_OK) {
      dliom_grid_destroy(*out);
      *out = nullptr;
      return s;
    }
  }
  return DLIOM_OK;
}


// ---- mapping::proto::Submap3D around two serialized grids (mapping/proto/submap.proto, transform/proto/transform.proto)
namespace {
void put_double_field(std::vector<uint8_t>* out, int field, double v) {
  if (v == 0.0) return;  // proto3 implicit presence, as the C++ runtime decides it (x != 0)
  put_varint(out, static_cast<uint64_t>(field) << 3 | 1);
  uint64_t bits;
  std::memcpy(&bits, &v, 8);
  for (int i = 0; i < 8; ++i) out->push_back(static_cast<uint8_t>(bits >> (8 * i)));
}
void put_message(std::vector<uint8_t>* out, int field, const std::vector<uint8_t>& payload) {
  put_varint(out, static_cast<uint64_t>(field) << 3 | 2);  // a present sub-message is emitted even when empty
  put_varint(out, payload.size());
  out->insert(out->end(), payload.begin(), payload.end());
}
void put_bytes(std::vector<uint8_t>* out, int field, const uint8_t* p, int64_t n) {
  put_varint(out, static_cast<uint64_t>(field) << 3 | 2);
  put_varint(out, static_cast<uint64_t>(n));
  out->insert(out->end(), p, p + n);
}
bool skip_field(const uint8_t*& p, const uint8_t* end, unsigned wire) {
  uint64_t v;
  switch (wire) {
    case 0: return get_varint(p, end, &v);
    case 1: if (end - p < 8) return false; p += 8; return true;
    case 2: if (!get_varint(p, end, &v) || static_cast<uint64_t>(end - p) < v) return false; p += v; return true;
    case 5: if (end - p < 4) return false; p += 4; return true;
    default: return false;
  }
}
// doubles of a Vector3d / Quaterniond message into dst[field - 1]
bool parse_doubles(const uint8_t* p, const uint8_t* end, double* dst, int max_field) {
  while (p < end) {
    uint64_t tag;
    if (!get_varint(p, end, &tag)) return false;
    const unsigned wire = static_cast<unsigned>(tag & 7), field = static_cast<unsigned>(tag >> 3);
    if (wire == 1 && field >= 1 && field <= static_cast<unsigned>(max_field)) {
      if (end - p < 8) return false;
      uint64_t bits = 0;
      for (int i = 0; i < 8; ++i) bits |= static_cast<uint64_t>(p[i]) << (8 * i);
      std::memcpy(&dst[field - 1], &bits, 8);
      p += 8;
    } else if (!skip_field(p, end, wire)) {
      return false;
    }
  }
  return true;
}
}  // namespace

extern "C" int dliom_submap3d_to_proto(const double local_pose7[7], int32_t num_range_data, int finished,
                                       const uint8_t* hi, int64_t hi_size, const uint8_t* lo, int64_t lo_size,
                                       int wrap_in_submap, uint8_t* buffer, int64_t capacity, int64_t* size) {
  if (local_pose7 == nullptr || size == nullptr || capacity < 0 || (hi != nullptr && hi_size < 0) ||
      (lo != nullptr && lo_size < 0))
    return DLIOM_ERR_INVALID_ARGUMENT;
  std::vector<uint8_t> translation, rotation, pose, msg;
  for (int i = 0; i < 3; ++i) put_double_field(&translation, i + 1, local_pose7[i]);  // Vector3d x, y, z
  put_double_field(&rotation, 1, local_pose7[4]);                                      // Quaterniond x, y, z, w
  put_double_field(&rotation, 2, local_pose7[5]);
  put_double_field(&rotation, 3, local_pose7[6]);
  put_double_field(&rotation, 4, local_pose7[3]);
  put_message(&pose, 1, translation);  // transform::ToProto(Rigid3d) sets both sub-messages
  put_message(&pose, 2, rotation);
  put_message(&msg, 1, pose);
  if (num_range_data != 0) {
    put_varint(&msg, 2u << 3 | 0);
    put_varint(&msg, static_cast<uint64_t>(static_cast<int64_t>(num_range_data)));  // int32: sign-extended to 64 bits
  }
  if (finished != 0) {
    put_varint(&msg, 3u << 3 | 0);
    msg.push_back(1);
  }
  if (hi != nullptr) put_bytes(&msg, 4, hi, hi_size);
  if (lo != nullptr) put_bytes(&msg, 5, lo, lo_size);
  std::vector<uint8_t> outer;
  if (wrap_in_submap != 0) {
    put_message(&outer, 2, msg);  // proto::Submap.submap_3d
    msg.swap(outer);
  }
  *size = static_cast<int64_t>(msg.size());
  if (buffer == nullptr) return DLIOM_OK;
  if (capacity < *size) return DLIOM_ERR_CAPACITY;
  std::memcpy(buffer, msg.data(), msg.size());
  return DLIOM_OK;
}

extern "C" int dliom_submap3d_from_proto(const uint8_t* buffer, int64_t size, int wrapped_in_submap, double local_pose7[7],
                                         int32_t* num_range_data, int* finished, int64_t* hi_offset, int64_t* hi_size,
                                         int64_t* lo_offset, int64_t* lo_size) {
  if (buffer == nullptr || size < 0 || local_pose7 == nullptr || num_range_data == nullptr || finished == nullptr ||
      hi_offset == nullptr || hi_size == nullptr || lo_offset == nullptr || lo_size == nullptr)
    return DLIOM_ERR_INVALID_ARGUMENT;
  const uint8_t* p = buffer;
  const uint8_t* end = buffer + size;
  if (wrapped_in_submap != 0) {  // proto::Submap: find field 2 (the last occurrence wins, as protobuf merges)
    const uint8_t* body = nullptr;
    uint64_t body_len = 0;
    while (p < end) {
      uint64_t tag, len;
      if (!get_varint(p, end, &tag)) return DLIOM_ERR_INVALID_ARGUMENT;
      if ((tag & 7) == 2 && (tag >> 3) == 2) {
        if (!get_varint(p, end, &len) || static_cast<uint64_t>(end - p) < len) return DLIOM_ERR_INVALID_ARGUMENT;
        body = p;
        body_len = len;
        p += len;
      } else if (!skip_field(p, end, static_cast<unsigned>(tag & 7))) {
        return DLIOM_ERR_INVALID_ARGUMENT;
      }
    }
    if (body == nullptr) return DLIOM_ERR_INVALID_ARGUMENT;  // CHECK(proto.has_submap_3d())
    p = body;
    end = body + body_len;
  }
  double t[3] = {0, 0, 0}, q[4] = {0, 0, 0, 0};  // proto3 defaults (x, y, z, w)
  *num_range_data = 0;
  *finished = 0;
  *hi_offset = *lo_offset = 0;
  *hi_size = *lo_size = -1;
  while (p < end) {
    uint64_t tag;
    if (!get_varint(p, end, &tag)) return DLIOM_ERR_INVALID_ARGUMENT;
    const unsigned wire = static_cast<unsigned>(tag & 7), field = static_cast<unsigned>(tag >> 3);
    if (wire == 2 && (field == 1 || field == 4 || field == 5)) {
      uint64_t len;
      if (!get_varint(p, end, &len) || static_cast<uint64_t>(end - p) < len) return DLIOM_ERR_INVALID_ARGUMENT;
      if (field == 4) {
        *hi_offset = p - buffer;
        *hi_size = static_cast<int64_t>(len);
      } else if (field == 5) {
        *lo_offset = p - buffer;
        *lo_size = static_cast<int64_t>(len);
      } else {  // Rigid3d: 1 translation, 2 rotation
        const uint8_t* r = p;
        const uint8_t* rend = p + len;
        while (r < rend) {
          uint64_t rtag, rlen;
          if (!get_varint(r, rend, &rtag)) return DLIOM_ERR_INVALID_ARGUMENT;
          if ((rtag & 7) == 2 && ((rtag >> 3) == 1 || (rtag >> 3) == 2)) {
            if (!get_varint(r, rend, &rlen) || static_cast<uint64_t>(rend - r) < rlen) return DLIOM_ERR_INVALID_ARGUMENT;
            const bool ok = (rtag >> 3) == 1 ? parse_doubles(r, r + rlen, t, 3) : parse_doubles(r, r + rlen, q, 4);
            if (!ok) return DLIOM_ERR_INVALID_ARGUMENT;
            r += rlen;
          } else if (!skip_field(r, rend, static_cast<unsigned>(rtag & 7))) {
            return DLIOM_ERR_INVALID_ARGUMENT;
          }
        }
      }
      p += len;
    } else if (wire == 0 && (field == 2 || field == 3)) {
      uint64_t v;
      if (!get_varint(p, end, &v)) return DLIOM_ERR_INVALID_ARGUMENT;
      if (field == 2) *num_range_data = static_cast<int32_t>(v);
      else *finished = v != 0 ? 1 : 0;
    } else if (!skip_field(p, end, wire)) {
      return DLIOM_ERR_INVALID_ARGUMENT;
    }
  }
  local_pose7[0] = t[0];
  local_pose7[1] = t[1];
  local_pose7[2] = t[2];
  local_pose7[3] = q[3];  // w
  local_pose7[4] = q[0];
  local_pose7[5] = q[1];
  local_pose7[6] = q[2];
  return DLIOM_OK;
}
