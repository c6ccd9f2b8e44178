// nmi.hh -- normalised mutual information of two covers (overlapping communities).
//
// The reference shells out to the external program /usr/local/bin/mutual (Lancichinetti, Fortunato &
// Kertesz's "mutual3") after every communities.txt it writes and appends that program's line
// "mutual3:\t<value>" to mutual.txt (src/linksampling.cc:839-852).  That program is not part of the
// reference tree; this is the published measure it computes (New J. Phys. 11 (2009) 033015, eq. B.10-B.14):
//   N(X|Y) = 1 - [H(X|Y)_norm + H(Y|X)_norm] / 2,
//   H(X|Y)_norm = mean_k  min_l* H(X_k|Y_l) / H(X_k),
// with the minimum over the l for which h(P11) + h(P00) >= h(P01) + h(P10), and H(X_k) when there is none.
// Pinned by the authors' own shipped run (tests/golden/ref_lfr_k28: communities.txt against the LFR
// ground truth gives the last line of their mutual.txt, 0.897372).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace svinet {

typedef std::vector<std::vector<uint32_t> > Cover;   // communities as lists of node ids

// one community per line, ids separated by white space (communities.txt, ground_truth.txt)
bool read_cover_lines(const std::string &path, Cover *out);
// "node <TAB> community community ..." per line (the LFR benchmark's community file; what -nmi takes,
// Network::load_ground_truth, src/network.cc:252-307); communities come out in ascending id order
bool read_cover_memberships(const std::string &path, Cover *out);
double lfk_nmi(const Cover &x, const Cover &y);

}  // namespace svinet
