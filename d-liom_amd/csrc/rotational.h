// RotationalScanMatcher pieces shared by fast_csm3d.hip and the host-only entry points
// (rotational_histogram.cc): mapping/internal/3d/scan_matching/rotational_scan_matcher.cc:125-194.
#ifndef DLIOM_CSRC_ROTATIONAL_H_
#define DLIOM_CSRC_ROTATIONAL_H_

#include <vector>

namespace dliom {
// RotateHistogram (:125-144): fractional bucket rotation with linear interpolation.
std::vector<float> rotate_histogram(const std::vector<float>& histogram, float angle);
// MatchHistograms (:146-157): dot product of the normalised histograms, Eigen's vectorised
// reduction order over dynamic float vectors restated.
float match_histograms(const std::vector<float>& submap_histogram, const std::vector<float>& scan_histogram);
}  // namespace dliom

#endif  // DLIOM_CSRC_ROTATIONAL_H_
