// ORACLE (test infrastructure, NOT product code).
//
// 15-bit occupancy codec of the reference, restated around one small
// "bounded float <-> [1,32767]" codec object instead of the reference's
// free functions.  Arithmetic (all in float) follows
//   mapping/probability_values.h:32-44   value = lround((clamp(f)-lo)*(32766/(hi-lo)))+1
//   mapping/probability_values.h:48-68   Odds, ProbabilityFromOdds, k{Min,Max}*
//   mapping/probability_values.h:80-82   unknown = 0, update marker = 1<<15
//   mapping/probability_values.cc:27-36  f = value*kScale + (lo - kScale)
//   mapping/probability_values.cc:38-71  65536-entry tables (repeated for marker)
//   mapping/probability_values.cc:73-101 odds-update lookup tables
//   mapping/probability_values.h:111-141 probability <-> correspondence-cost values
#ifndef ORACLE_OM_PROBABILITY_VALUES_H_
#define ORACLE_OM_PROBABILITY_VALUES_H_

#include <vector>

#include "om_math.h"

namespace oracle {

constexpr float kMinProbability = 0.1f;
constexpr float kMaxProbability = 1.f - kMinProbability;
constexpr float kMinCorrespondenceCost = 1.f - kMaxProbability;
constexpr float kMaxCorrespondenceCost = 1.f - kMinProbability;
constexpr uint16 kUnknownProbabilityValue = 0;
constexpr uint16 kUnknownCorrespondenceValue = 0;
constexpr uint16 kUpdateMarker = 1u << 15;

inline float Odds(float p) { return p / (1.f - p); }
inline float ProbabilityFromOdds(float odds) { return odds / (odds + 1.f); }
inline float ProbabilityToCorrespondenceCost(float p) { return 1.f - p; }
inline float CorrespondenceCostToProbability(float c) { return 1.f - c; }
inline float ClampProbability(float p) {
  return Clamp(p, kMinProbability, kMaxProbability);
}

// One codec per encoded quantity (probability, correspondence cost).
struct BoundedCodec {
  float lo, hi;
  float unknown_result;  // what value 0 decodes to

  uint16 Encode(float f) const {  // probability_values.h:32-44
    const int v = RoundToInt((Clamp(f, lo, hi) - lo) * (32766.f / (hi - lo))) + 1;
    return static_cast<uint16>(v);
  }
  float DecodeSlow(uint16 v) const {  // probability_values.cc:27-36
    if (v == 0) return unknown_result;
    const float kScale = (hi - lo) / 32766.f;
    return v * kScale + (lo - kScale);
  }
  std::vector<float> Table() const {  // probability_values.cc:38-51
    std::vector<float> t(65536);
    for (int v = 0; v != 32768; ++v) {
      t[v] = DecodeSlow(static_cast<uint16>(v));
      t[v + 32768] = t[v];  // same entry again for marker-carrying values
    }
    return t;
  }
};

inline const BoundedCodec& ProbabilityCodec() {
  static const BoundedCodec c{kMinProbability, kMaxProbability, kMinProbability};
  return c;
}
inline const BoundedCodec& CorrespondenceCostCodec() {
  static const BoundedCodec c{kMinCorrespondenceCost, kMaxCorrespondenceCost,
                              kMaxCorrespondenceCost};
  return c;
}
inline const std::vector<float>& ValueToProbabilityTable() {
  static const std::vector<float> t = ProbabilityCodec().Table();
  return t;
}
inline const std::vector<float>& ValueToCorrespondenceCostTable() {
  static const std::vector<float> t = CorrespondenceCostCodec().Table();
  return t;
}

inline uint16 ProbabilityToValue(float p) { return ProbabilityCodec().Encode(p); }
inline uint16 CorrespondenceCostToValue(float c) {
  return CorrespondenceCostCodec().Encode(c);
}
inline float ValueToProbability(uint16 v) { return ValueToProbabilityTable()[v]; }
inline float ValueToCorrespondenceCost(uint16 v) {
  return ValueToCorrespondenceCostTable()[v];
}

// probability_values.h:111-141.  NB the reference tests `> kUpdateMarker`
// (strictly), so the bare marker 32768 is passed through the tables as is.
inline uint16 ProbabilityValueToCorrespondenceCostValue(uint16 v) {
  if (v == kUnknownProbabilityValue) return kUnknownCorrespondenceValue;
  const bool carry = v > kUpdateMarker;
  if (carry) v -= kUpdateMarker;
  uint16 r = CorrespondenceCostToValue(
      ProbabilityToCorrespondenceCost(ValueToProbability(v)));
  if (carry) r += kUpdateMarker;
  return r;
}
inline uint16 CorrespondenceCostValueToProbabilityValue(uint16 v) {
  if (v == kUnknownCorrespondenceValue) return kUnknownProbabilityValue;
  const bool carry = v > kUpdateMarker;
  if (carry) v -= kUpdateMarker;
  uint16 r = ProbabilityToValue(
      CorrespondenceCostToProbability(ValueToCorrespondenceCost(v)));
  if (carry) r += kUpdateMarker;
  return r;
}

// probability_values.cc:73-83: entry 0 is "first observation", entries
// 1..32767 multiply the stored odds; every output carries the update marker.
inline std::vector<uint16> ComputeLookupTableToApplyOdds(float odds) {
  std::vector<uint16> t(32768);
  t[0] = ProbabilityToValue(ProbabilityFromOdds(odds)) + kUpdateMarker;
  const std::vector<float>& p = ValueToProbabilityTable();
  for (int cell = 1; cell != 32768; ++cell) {
    t[cell] = ProbabilityToValue(ProbabilityFromOdds(odds * Odds(p[cell]))) +
              kUpdateMarker;
  }
  return t;
}

// probability_values.cc:85-101 (2D ProbabilityGrid stores correspondence cost)
inline std::vector<uint16> ComputeLookupTableToApplyCorrespondenceCostOdds(
    float odds) {
  std::vector<uint16> t(32768);
  t[0] = CorrespondenceCostToValue(
             ProbabilityToCorrespondenceCost(ProbabilityFromOdds(odds))) +
         kUpdateMarker;
  const std::vector<float>& c = ValueToCorrespondenceCostTable();
  for (int cell = 1; cell != 32768; ++cell) {
    const float p = CorrespondenceCostToProbability(c[cell]);
    t[cell] = CorrespondenceCostToValue(ProbabilityToCorrespondenceCost(
                  ProbabilityFromOdds(odds * Odds(p)))) +
              kUpdateMarker;
  }
  return t;
}

}  // namespace oracle

#endif  // ORACLE_OM_PROBABILITY_VALUES_H_
