"""CPU restatement of the bookkeeping of FeatureTracker::track - numpy / plain Python.

TEST INFRASTRUCTURE ONLY (imported by tests/ and bench.py's cpu leg; never by dynosam_amd/).

Follows, line by line where the logic is order dependent:
  dynosam/src/frontend/vision/FeatureTracker.cc
      :73-192    track             composition: objectDetection -> static track || dynamic track -> Frame
      :339-498   trackDynamic      (the per-feature propagation itself is oracle/flow_oracle.py::track_dynamic)
      :864-1012  sampleDynamic     candidates = pixels with detection mask set, label to sample, non-zero flow, inside the
                                   shrunken image; per object ANMS (RangeTree) to max_features - num_track; age 0, new ids
      :1014-1147 requiresSampling  new object | > 80 % of the tracks about to expire | < min_dynamic_tracks | IoU(detection
                                   box, box of the tracks) < min_dynamic_mask_iou
  dynosam/src/frontend/anms/anms.cc:278-361   anms::RangeTree (brute-force square queries instead of the range tree)
  dynosam/src/frontend/anms/NonMaximumSupression.cc:33-56   the response sort in front of it
  dynosam/src/frontend/vision/StaticFeatureTracker.cc
      :244-300   trackStatic       first frame -> detectFeatures, else trackPoints           (track_static_frame below)
      :420-637   trackPoints       LK + flow-back (klt_oracle), RANSAC homography (ransac_oracle), usable / age tests,
                                   outliers, top-up below min_features_per_frame
      :330-418   detectFeatures    detection mask (caller mask & background & discs), SparseFeatureDetector::detect
                                   (FeatureDetector.cc:186-241: CLAHE -> corners -> ANMS -> cornerSubPix: clahe_oracle, gftt_oracle,
                                   anms_range_tree above, subpix_oracle), contained / shrunken / background tests, new ids
      :1212-1358 propogateMask     an object whose previous tracks now land on background gets its previous mask warped forward
                                   by the dense flow into the current mask                      (propogate_mask below)
Undefined in the reference and fixed here (and in the product): the candidate order inside an object (a tbb::parallel_for over
rows fills the vectors; all responses are equal) = row-major; num_ret <= 0 -> nothing, num_ret == 1 -> the first keypoint (the
reference divides by num_ret - 1 / num_ret).  PARITY UNPINNED against the binary (the reference has no test for any of this).
"""
from __future__ import annotations

import numpy as np


def range_tree_box(tx, ty, minx, maxx, miny, maxy):
    """rangetree<u16, u16>::search(minx, maxx, miny, maxy) as a predicate on the tree coordinates tx, ty (the keypoints' (u16) truncations):
    the int corners become u16 at the call, a reversed pair is swapped, both ends are inclusive.  Pinned against the reference's own
    ranget.h compiled into oracle/_ref (tests/test_ref_anms_structs.py)."""
    x0, x1, y0, y1 = minx & 0xFFFF, maxx & 0xFFFF, miny & 0xFFFF, maxy & 0xFFFF
    if x1 < x0:
        x0, x1 = x1, x0
    if y1 < y0:
        y0, y1 = y1, y0
    return (tx >= x0) & (tx <= x1) & (ty >= y0) & (ty <= y1)


def kdtree_disc(px, py, qx, qy, radius):
    """nanoflann radiusSearch(query, radius * radius) over integer points as a predicate: squared distance STRICTLY below the squared
    radius.  Pinned against the reference's own nanoflann.hpp compiled into oracle/_ref (tests/test_ref_anms_structs.py)."""
    return ((px - qx) ** 2 + (py - qy) ** 2) < radius * radius


def anms_range_tree(xy: np.ndarray, num_ret: int, tolerance: float, cols: int, rows: int, box_query=None) -> np.ndarray:
    """indices (into xy) of the kept keypoints, in selection order.  box_query(minx, maxx, miny, maxy) -> indices or mask of the hit
    keypoints replaces the built-in predicate (the tests plug the reference's compiled range tree in there)."""
    xy = np.asarray(xy, np.float32).reshape(-1, 2)
    n = len(xy)
    K = int(num_ret)
    if n == 0 or K <= 0:
        return np.zeros(0, np.int64)
    if K == 1:
        return np.zeros(1, np.int64)
    exp1 = rows + cols + 2 * K
    exp2 = 4 * cols + 4 * K + 4 * rows * K + rows * rows + cols * cols - 2 * rows * cols + 4 * rows * cols * K
    exp3 = np.sqrt(float(exp2))
    exp4 = float(K - 1)
    c_round = lambda v: float(np.sign(v) * np.floor(abs(v) + 0.5))     # C round(): half away from zero
    sol1, sol2 = -c_round((exp1 + exp3) / exp4), -c_round((exp1 - exp3) / exp4)
    high = int(sol1 if sol1 > sol2 else sol2)
    low = int(np.floor(np.sqrt(float(n) / K)))
    tol = np.float32(tolerance)
    Kf = np.float32(K)
    kmin = int(c_round(float(Kf - Kf * tol)))
    kmax = int(c_round(float(Kf + Kf * tol)))
    tx = xy[:, 0].astype(np.int64) & 0xFFFF      # tree coordinates: (u16) truncation
    ty = xy[:, 1].astype(np.int64) & 0xFFFF
    result, prevwidth = [], -1
    while True:
        width = low + int((high - low) / 2)       # C integer division truncates toward zero
        if width == prevwidth or low > high:
            return np.array(result, np.int64)
        result = []
        included = np.ones(n, bool)
        for i in range(n):
            if not included[i]:
                continue
            included[i] = False
            result.append(i)
            w32 = np.float32(width)
            minx, maxx = int(xy[i, 0] - w32), int(xy[i, 0] + w32)
            miny, maxy = int(xy[i, 1] - w32), int(xy[i, 1] + w32)
            minx, miny = max(minx, 0), max(miny, 0)
            if box_query is None:
                included &= ~range_tree_box(tx, ty, minx, maxx, miny, maxy)
            else:
                included[box_query(minx, maxx, miny, maxy)] = False
        if kmin <= len(result) <= kmax:
            return np.array(result, np.int64)
        if len(result) < kmin:
            high = width - 1
        else:
            low = width + 1
        prevwidth = width


def _c_round(v):
    """C round(): half away from zero"""
    return float(np.sign(v) * np.floor(abs(v) + 0.5))


def _k_range(K, tolerance):
    Kf, tol = np.float32(K), np.float32(tolerance)
    return int(_c_round(float(Kf - Kf * tol))), int(_c_round(float(Kf + Kf * tol)))


def anms_top_n(xy, num_ret):
    """anms::TopN (anms.cc:67-78): the first num_ret keypoints of the list as it is handed over"""
    n = len(xy)
    return np.arange(n if num_ret > n else max(num_ret, 0), dtype=np.int64)


def anms_brown(xy, num_ret):
    """anms::BrownANMS (anms.cc:80-107): every keypoint's distance to the nearest EARLIER one (fp32), the num_ret largest radii"""
    xy = np.asarray(xy, np.float32).reshape(-1, 2)
    n = len(xy)
    if num_ret > n:
        return np.arange(n, dtype=np.int64)
    rad = np.full(n, np.finfo(np.float32).max, np.float32)
    for i in range(1, n):
        e1, e2 = xy[:i, 0] - xy[i, 0], xy[:i, 1] - xy[i, 1]
        rad[i] = np.sqrt(e1 * e1 + e2 * e2, dtype=np.float32).min()
    # sort(results.begin(), results.end(), sort_pred()) on (radius, index) pairs with `left.first > right.first`: std::sort, so keypoints of EQUAL
    # radius (integer pixel positions make those common) come out in libstdc++'s order - sorting the pairs moves them exactly as sorting indices does
    order = std_sort_indices([float(r) for r in rad], comp=lambda x, y: x > y)
    return order[:max(num_ret, 0)].astype(np.int64)


def _grid_cover(xy, n, cols, rows, c, reach, disc):
    """the covering pass Sdc and Ssc share: cells of side c, a taken keypoint covers the cells within `reach` cells (a disc of that
    radius in cell units when disc, else the square)"""
    ncc, ncr = int(np.floor(cols / c)), int(np.floor(rows / c))
    covered = np.zeros((ncr + 1, ncc + 1), bool)
    result = []
    fl = int(np.floor(reach))
    for i in range(n):
        row, col = int(np.floor(float(xy[i, 1]) / c)), int(np.floor(float(xy[i, 0]) / c))
        if covered[row, col]:
            continue
        result.append(i)
        r0, r1, c0, c1 = max(row - fl, 0), min(row + fl, ncr), max(col - fl, 0), min(col + fl, ncc)
        if disc:
            rr, cc = np.mgrid[r0:r1 + 1, c0:c1 + 1]
            covered[r0:r1 + 1, c0:c1 + 1] |= np.sqrt(((rr - row) ** 2 + (cc - col) ** 2).astype(np.float64)) <= reach
        else:
            covered[r0:r1 + 1, c0:c1 + 1] = True
    return result


def anms_sdc(xy, num_ret, tolerance, cols, rows):
    """anms::Sdc (anms.cc:109-186): suppression via disc covering, binary search over the radius in [1, cols]; as there, prevradius is
    never updated - the search ends when low passes high"""
    xy = np.asarray(xy, np.float32).reshape(-1, 2)
    n = len(xy)
    low, high = 1, cols
    kmin, kmax = _k_range(num_ret, tolerance)
    result = []
    while True:
        radius = low + int((high - low) / 2)
        if radius == -1 or low > high:
            return np.array(result, np.int64)
        c = 0.25 * radius / np.sqrt(2.0)
        result = _grid_cover(xy, n, cols, rows, c, radius / c, True)
        if kmin <= len(result) <= kmax:
            return np.array(result, np.int64)
        if len(result) < kmin:
            high = radius - 1
        else:
            low = radius + 1


def _search_range(n, K, cols, rows):
    exp1 = rows + cols + 2 * K
    exp2 = 4 * cols + 4 * K + 4 * rows * K + rows * rows + cols * cols - 2 * rows * cols + 4 * rows * cols * K
    exp3, exp4 = np.sqrt(float(exp2)), float(K - 1)
    sol1, sol2 = -_c_round((exp1 + exp3) / exp4), -_c_round((exp1 - exp3) / exp4)
    return int(np.floor(np.sqrt(float(n) / K))), int(sol1 if sol1 > sol2 else sol2)


def anms_kdtree(xy, num_ret, tolerance, cols, rows, disc_query=None):
    """anms::KdTree (anms.cc:188-276): a taken keypoint excludes every keypoint whose truncated integer position lies closer than
    the radius (nanoflann radiusSearch on squared distances, strictly smaller; the tree only answers that question)"""
    xy = np.asarray(xy, np.float32).reshape(-1, 2)
    n = len(xy)
    low, high = _search_range(n, num_ret, cols, rows)
    kmin, kmax = _k_range(num_ret, tolerance)
    px, py = xy[:, 0].astype(np.int64), xy[:, 1].astype(np.int64)
    result, prev = [], -1
    while True:
        radius = low + int((high - low) / 2)
        if radius == prev or low > high:
            return np.array(result, np.int64)
        result = []
        included = np.ones(n, bool)
        for i in range(n):
            if not included[i]:
                continue
            included[i] = False
            result.append(i)
            if disc_query is None:
                included &= ~kdtree_disc(px, py, px[i], py[i], radius)
            else:
                included[disc_query(xy[i, 0], xy[i, 1], radius)] = False
        if kmin <= len(result) <= kmax:
            return np.array(result, np.int64)
        if len(result) < kmin:
            high = radius - 1
        else:
            low = radius + 1
        prev = radius


def anms_ssc(xy, num_ret, tolerance, cols, rows):
    """anms::Ssc (anms.cc:364-475): suppression via square covering; cell side = width / 2 in INTEGER division (:425), the search also
    ends at low >= high (:415-421).  A width of 1 makes the cell side 0 (the reference divides by it): ValueError here"""
    xy = np.asarray(xy, np.float32).reshape(-1, 2)
    n = len(xy)
    low, high = _search_range(n, num_ret, cols, rows)
    kmin, kmax = _k_range(num_ret, tolerance)
    result, prev = [], -1
    while True:
        width = low + int((high - low) / 2)
        if width == prev or low >= high:
            return np.array(result, np.int64)
        c = float(int(width / 2))
        if c <= 0:
            raise ValueError("anms::Ssc: cell side 0")
        result = _grid_cover(xy, n, cols, rows, c, width / c, False)
        if kmin <= len(result) <= kmax:
            return np.array(result, np.int64)
        if len(result) < kmin:
            high = width - 1
        else:
            low = width + 1
        prev = width


def anms_binning(xy, num_ret, cols, rows, nr_horizontal_bins, nr_vertical_bins, binning_mask):
    """AdaptiveNonMaximumSuppression::binning (NonMaximumSupression.cc:117-159): per active bin the first round(num_ret / active bins)
    keypoints of the list"""
    xy = np.asarray(xy, np.float32).reshape(-1, 2)
    n = len(xy)
    if num_ret > n:
        return np.arange(n, dtype=np.int64)
    mask = np.asarray(binning_mask, np.float64).reshape(nr_vertical_bins, nr_horizontal_bins)
    bin_r, bin_c = np.float32(rows) / np.float32(nr_vertical_bins), np.float32(cols) / np.float32(nr_horizontal_bins)
    active = np.float32(mask.sum())
    if not active > 0:
        raise ValueError("binning: no active bin")
    per_bin = int(_c_round(float(np.float32(num_ret) / active)))
    cnt = np.zeros((nr_vertical_bins, nr_horizontal_bins), np.int64)
    out = []
    for i in range(n):
        r, c = int(xy[i, 1] / bin_r), int(xy[i, 0] / bin_c)
        if mask[r, c] == 1 and cnt[r, c] < per_bin:
            out.append(i)
            cnt[r, c] += 1
    return np.array(out, np.int64)


def std_sort_indices(keys, comp=None):
    """libstdc++'s std::sort (bits/stl_algo.h: introsort - median-of-three to the front, unguarded Hoare partition, recursion on the right part, depth
    limit 2 lg n, then one insertion-sort pass with the first 16 guarded) applied to the index array 0..n-1 with the comparator keys[a] < keys[b], as
    cv::sortIdx's generic path does (modules/core/src/matrix_operations.cpp: sortIdx_ -> std::sort(iptr, iptr + len, LessThanIdx<T>(ptr))).  The sort is
    not stable: WHERE equal keys end up is part of the reference's behaviour on OpenCV builds without IPP.  Pinned against g++'s own std::sort by
    tests/test_anms_types.py.  (The heap-sort fallback behind the depth limit is not restated: it raises.)  comp(x, y): the strict weak order on
    the keys, default x < y."""
    k = list(keys)
    a = list(range(len(k)))
    less = (lambda x, y: k[x] < k[y]) if comp is None else (lambda x, y: comp(k[x], k[y]))

    def unguarded_linear_insert(last):
        val = a[last]
        nxt = last - 1
        while less(val, a[nxt]):
            a[last] = a[nxt]
            last = nxt
            nxt -= 1
        a[last] = val

    def insertion_sort(first, last):
        if first == last:
            return
        for i in range(first + 1, last):
            if less(a[i], a[first]):
                val = a[i]
                a[first + 1:i + 1] = a[first:i]
                a[first] = val
            else:
                unguarded_linear_insert(i)

    def move_median_to_first(result, x, y, z):
        if less(a[x], a[y]):
            pick = y if less(a[y], a[z]) else (z if less(a[x], a[z]) else x)
        elif less(a[x], a[z]):
            pick = x
        elif less(a[y], a[z]):
            pick = z
        else:
            pick = y
        a[result], a[pick] = a[pick], a[result]

    def unguarded_partition(first, last, pivot):
        while True:
            while less(a[first], a[pivot]):
                first += 1
            last -= 1
            while less(a[pivot], a[last]):
                last -= 1
            if not first < last:
                return first
            a[first], a[last] = a[last], a[first]
            first += 1

    def introsort_loop(first, last, depth):
        while last - first > 16:
            if depth == 0:
                raise NotImplementedError("std::sort's heap-sort fallback")
            depth -= 1
            mid = first + (last - first) // 2
            move_median_to_first(first, first + 1, mid, last - 1)
            cut = unguarded_partition(first + 1, last, first)
            introsort_loop(cut, last, depth)
            last = cut

    n = len(a)
    if n:
        introsort_loop(0, n, 2 * (n.bit_length() - 1))
        if n > 16:
            insertion_sort(0, 16)
            for i in range(16, n):
                unguarded_linear_insert(i)
        else:
            insertion_sort(0, n)
    return np.array(a, np.int64)


def sort_idx_descending(keys, std_sort=False):
    """cv::sortIdx(keys, SORT_DESCENDING) on the row of (int) responses: IPP's radix sort where OpenCV is built with IPP (x86 defaults; equal keys keep
    their order) or - std_sort - the generic path: std::sort ascending, then the index array reversed"""
    k = np.asarray(keys).astype(np.int64)
    if not std_sort:
        return np.argsort(-k, kind="stable")
    return std_sort_indices([int(v) for v in k])[::-1].copy()


ANMS_STD_SORT = 0x100     # flag on the ANMS type: the response sort as OpenCV's generic cv::sortIdx performs it (builds without IPP, e.g. docker/Dockerfile.l4t_jetpack6)
ANMS_TYPES = {"TopN": 0, "BrownANMS": 1, "SDC": 2, "KdTree": 3, "RangeTree": 4, "Ssc": 5, "Binning": 6}      # AnmsAlgorithmType (NonMaximumSuppression.h:49-57)


def suppress_non_max(xy, response, num_ret, tolerance, cols, rows, anms_type=4, nr_horizontal_bins=5, nr_vertical_bins=5, binning_mask=None):
    """AdaptiveNonMaximumSuppression::suppressNonMax (NonMaximumSupression.cc:33-115): indices into xy of the kept keypoints, in the order
    they are handed back.  The list is sorted by (int)response, descending (equal responses in their order - or, with the ANMS_STD_SORT flag on the
    type, where std::sort leaves them; response None = all equal, cv::GFTTDetector's keypoints) - except for TopN and BrownANMS, which the reference
    hands the UNSORTED list (:65,71)."""
    xy = np.asarray(xy, np.float32).reshape(-1, 2)
    n = len(xy)
    if n == 0:
        return np.zeros(0, np.int64)
    std_sort, anms_type = bool(anms_type & ANMS_STD_SORT), anms_type & 0xFF
    order = sort_idx_descending(np.zeros(n, np.int64) if response is None else np.asarray(response).astype(np.int64), std_sort)
    if anms_type == 0:
        return anms_top_n(xy, num_ret)
    if anms_type == 1:
        return anms_brown(xy, num_ret)
    s = xy[order]
    if anms_type in (2, 3, 5) and num_ret <= 0:
        return np.zeros(0, np.int64)
    if anms_type in (3, 5) and num_ret == 1:
        return np.zeros(0, np.int64)          # the search range divides by num_ret - 1: (int)(-inf) = INT_MIN on x86, the search ends at once
    if anms_type == 2:
        k = anms_sdc(s, num_ret, tolerance, cols, rows)
    elif anms_type == 3:
        k = anms_kdtree(s, num_ret, tolerance, cols, rows)
    elif anms_type == 4:
        k = anms_range_tree(s, num_ret, tolerance, cols, rows)
    elif anms_type == 5:
        k = anms_ssc(s, num_ret, tolerance, cols, rows)
    elif anms_type == 6:
        k = anms_binning(s, num_ret, cols, rows, nr_horizontal_bins, nr_vertical_bins, binning_mask)
    else:
        raise ValueError(anms_type)
    return order[k]


def determine_outlier_ids(inliers, tracklets):
    """determineOutlierIds (dynosam/src/frontend/vision/VisionTools.cc:744-764): both lists sorted, outliers = tracklets \\ inliers by
    std::set_difference - ascending ids whatever the order of the inputs.  Pinned by dynosam/test/test_tools.cc:41-68."""
    a, b = sorted(int(t) for t in tracklets), sorted(int(t) for t in inliers)
    out, j = [], 0
    for t in a:                                   # std::set_difference on sorted ranges (multiset semantics)
        while j < len(b) and b[j] < t:
            j += 1
        if j < len(b) and b[j] == t:
            j += 1
        else:
            out.append(t)
    return np.array(out, np.int64)


def within_shrunken(x, y, w, h, shrink_row, shrink_col):
    """FeatureTrackerBase::isWithinShrunkenImage on (col, row) = static_cast<int>(kp)"""
    c, r = np.asarray(x).astype(np.int64), np.asarray(y).astype(np.int64)
    return (r > shrink_row) & (r < h - shrink_row) & (c > shrink_col) & (c < w - shrink_col)


def propogate_mask(prev_object_id, prev_predicted_kp, prev_mask, prev_flow, cur_mask, shrink_row=0, shrink_col=0, min_points=150):
    """FeatureTracker::propogateMask (FeatureTracker.cc:1212-1358), called between objectDetection and the tracks when
    use_propogate_mask (:107-110).  prev_object_id [n], prev_predicted_kp [n, 2]: the usable dynamic features of frame k-1 (in
    container order); prev_mask [H, W] int, prev_flow [H, W, 2] float32: frame k-1's mask and its flow to frame k; cur_mask: frame k's
    mask.  Returns (mask of frame k after the propagation, labels that were propagated).
      per label of frame k-1's features, ascending (:1233-1236): the labels of the CURRENT mask at the features' predicted keypoints
      (static_cast<int>, strictly inside the image: :1273-1276); fewer than 150 of them -> skipped (:1281); the most frequent label -
      ties go to the smallest one: the counts are sorted by a std::sort over < 16 map entries in key order, which libstdc++ runs as
      an insertion sort that leaves equal elements in place (:1289-1309) - must be 0 (:1322); then every pixel of the PREVIOUS mask
      with that label and a flow whose two components are both non-zero (:1332) is moved by its flow and, if the target
      (static_cast<int>) is inside the shrunken image (:1341) and the un-truncated target strictly inside the image (:1345-1346),
      stamps the label into the current mask.  Labels are processed one after the other on the SAME mask: a later label sees (and may
      overwrite) what an earlier one stamped."""
    prev_object_id = np.asarray(prev_object_id, np.int64)
    prev_predicted_kp = np.asarray(prev_predicted_kp, np.float64).reshape(-1, 2)
    prev_mask = np.asarray(prev_mask)
    flow = np.asarray(prev_flow, np.float32)
    out = np.array(cur_mask, dtype=np.int32, copy=True)
    h, w = out.shape
    done = []
    for lab in sorted(set(int(o) for o in prev_object_id)):
        pk = prev_predicted_kp[prev_object_id == lab]
        u, v = pk[:, 0].astype(np.int64), pk[:, 1].astype(np.int64)          # functional_keypoint::u / v: truncation
        ok = (u < w) & (u > 0) & (v < h) & (v > 0)
        votes = out[v[ok], u[ok]]
        if len(votes) < min_points:
            continue
        labels, counts = np.unique(votes, return_counts=True)                # ascending labels; argmax takes the first maximum
        if int(labels[int(np.argmax(counts))]) != 0:
            continue
        rows, cols = np.nonzero(prev_mask == lab)
        fx, fy = flow[rows, cols, 0].astype(np.float64), flow[rows, cols, 1].astype(np.float64)
        px, py = cols.astype(np.float64) + fx, rows.astype(np.float64) + fy
        keep = (fx != 0) & (fy != 0) & within_shrunken(px, py, w, h, shrink_row, shrink_col) & (px < w) & (px > 0) & (py < h) & (py > 0)
        out[py[keep].astype(np.int64), px[keep].astype(np.int64)] = lab
        done.append(lab)
    return out, done


def sample_dynamic(motion_mask, flow, detection_mask, objects, n_needed, shrink_row=0, shrink_col=0, tolerance=0.01, next_tracklet_id=0):
    """FeatureTracker::sampleDynamic.  returns dict(label, tracklet_id, kp, flow, predicted_kp, n_candidates, n_sampled, n_zero_flow,
    next_tracklet_id); objects in the order given, ANMS order inside an object."""
    h, w = motion_mask.shape
    det = np.ones((h, w), bool) if detection_mask is None else (np.asarray(detection_mask) != 0)
    fx, fy = flow[..., 0], flow[..., 1]
    out = dict(label=[], tracklet_id=[], kp=[], flow=[], predicted_kp=[], n_candidates=[], n_sampled=[], n_zero_flow=[])
    ys_all, xs_all = np.mgrid[0:h, 0:w]
    inside = within_shrunken(xs_all, ys_all, w, h, shrink_row, shrink_col)
    tid = int(next_tracklet_id)
    for obj, need in zip(objects, n_needed):
        sel = det & (motion_mask == obj) & (obj != 0)
        zero = sel & ((fx == 0) | (fy == 0))
        cand = sel & ~zero & inside
        ys, xs = np.nonzero(cand)                    # row-major
        out["n_candidates"].append(len(xs)); out["n_zero_flow"].append(int(zero.sum()))
        if len(xs) == 0:
            out["n_sampled"].append(0)
            continue
        keep = anms_range_tree(np.stack([xs, ys], -1).astype(np.float32), int(need), tolerance, w, h)
        out["n_sampled"].append(len(keep))
        for i in keep:
            x, y = int(xs[i]), int(ys[i])
            f = (float(fx[y, x]), float(fy[y, x]))
            out["label"].append(int(obj)); out["tracklet_id"].append(tid); tid += 1
            out["kp"].append((float(x), float(y))); out["flow"].append(f); out["predicted_kp"].append((x + f[0], y + f[1]))
    res = {k: np.array(v) for k, v in out.items()}
    for k in ("kp", "flow", "predicted_kp"):
        res[k] = res[k].reshape(-1, 2)
    res["next_tracklet_id"] = tid
    return res


def bounding_rect(kp):
    """cv::boundingRect of float points (toOpenCV keypoints): [floor(min), floor(max)] inclusive -> (x, y, w, h) [recalled]"""
    x0, y0 = int(np.floor(kp[:, 0].min())), int(np.floor(kp[:, 1].min()))
    x1, y1 = int(np.floor(kp[:, 0].max())), int(np.floor(kp[:, 1].max()))
    return (x0, y0, x1 - x0 + 1, y1 - y0 + 1)


def iou(a, b):
    """utils::calculateIoU of two cv::Rect (x, y, w, h): intersection / union of the areas"""
    ax0, ay0, ax1, ay1 = a[0], a[1], a[0] + a[2], a[1] + a[3]
    bx0, by0, bx1, by1 = b[0], b[1], b[0] + b[2], b[1] + b[3]
    iw, ih = max(0, min(ax1, bx1) - max(ax0, bx0)), max(0, min(ay1, by1) - max(ay0, by0))
    inter = iw * ih
    union = a[2] * a[3] + b[2] * b[3] - inter
    return inter / union if union > 0 else 0.0


def requires_sampling(detected_objects, inner_boxes, tracked, known_objects, max_age=25, age_buffer=3, min_tracks=20, min_iou=0.4):
    """FeatureTracker::requiresSampling.  tracked: {object: dict(age [n], kp [n,2])} of this frame's tracked features;
    known_objects: labels with a PerObjectStatus in info_ (created by the tracking loop).  returns (objects to sample - ascending,
    as the std::set iterates -, {object: reasons})"""
    expiry = max_age - max(3, age_buffer)
    out, why = [], {}
    for obj, box in zip(detected_objects, inner_boxes):
        if obj in known_objects:
            if obj not in tracked or len(tracked[obj]["age"]) == 0:
                continue                                     # "found in mask and info but missing tracked features. Skipping"
            ages, kp = np.asarray(tracked[obj]["age"]), np.asarray(tracked[obj]["kp"], np.float64).reshape(-1, 2)
            n = len(ages)
            many_old = float((ages > expiry).sum()) / float(n) > 0.8
            too_few = n < min_tracks
            small = iou(tuple(box), bounding_rect(kp)) < min_iou
            if many_old or too_few or small:
                out.append(int(obj)); why[int(obj)] = dict(many_old=many_old, too_few=too_few, small_iou=small, new=False)
        else:
            out.append(int(obj)); why[int(obj)] = dict(many_old=False, too_few=False, small_iou=False, new=True)
    return sorted(out), why


def _status():
    return dict(num_previous_track=0, num_track=0, num_sampled=0, num_zero_flow=0, num_outside_shrunken_image=0,
                num_tracked_with_background_label=0, num_tracked_with_different_label=0, object_new=False, object_resampled=False)


def track_dynamic_frame(prev_dynamic, motion_mask, flow, boundary, next_tracklet_id, max_features=50, max_age=25, age_buffer=3, min_tracks=20,
                        min_iou=0.3, min_distance=2, shrink_row=0, shrink_col=0):
    """FeatureTracker::trackDynamic (:339-498) for one frame: propagation of the previous dynamic features (flow_oracle.track_dynamic),
    info_ bookkeeping, requiresSampling, sampleDynamic.  prev_dynamic: None or dict(tracklet_id, predicted_kp, age, object_id);
    boundary: dict(boundary_mask, objects, inner_boxes) of this frame's object mask; flow: the k -> k+1 flow image [H, W, 2] f32.
    returns (dict of the frame's dynamic features, objects sampled, per-object status, next tracklet id)"""
    from . import flow_oracle as FO
    status, tracked = {}, {}
    det = boundary["boundary_mask"]
    feats = dict(tracklet_id=[], kp=[], age=[], object_id=[], flow=[], predicted_kp=[])
    tid = int(next_tracklet_id)
    if prev_dynamic is not None and len(prev_dynamic["tracklet_id"]):
        r = FO.track_dynamic(prev_dynamic["predicted_kp"], prev_dynamic["object_id"], prev_dynamic["age"], prev_dynamic["tracklet_id"], flow, motion_mask,
                             detection_mask=boundary["boundary_mask"], shrink_row=shrink_row, shrink_col=shrink_col, max_age=max_age,
                             min_distance=min_distance, next_tracklet_id=tid)
        tid, det = r["next_tracklet_id"], r["detection_mask"]
        per_obj = {}
        for i, code in enumerate(r["code"]):
            if code == FO.MASKED_OUT:
                continue
            lab = int(r["label"][i])
            s = status.setdefault(lab, _status())
            s["num_previous_track"] += 1
            if lab == 0:
                s["num_tracked_with_background_label"] += 1
            if lab != int(prev_dynamic["object_id"][i]):
                s["num_tracked_with_different_label"] += 1
            if code == FO.OUTSIDE_SHRUNKEN:
                s["num_outside_shrunken_image"] += 1
            elif code == FO.ZERO_FLOW:
                s["num_zero_flow"] += 1
            elif code == FO.KEPT:
                s["num_track"] += 1
                per_obj.setdefault(lab, []).append(i)
        for lab in sorted(per_obj):                     # gtsam::FastMap iterates in ascending label order (:487-489)
            for i in per_obj[lab]:
                feats["tracklet_id"].append(int(r["new_tracklet_id"][i])); feats["kp"].append(tuple(prev_dynamic["predicted_kp"][i]))
                feats["age"].append(int(r["new_age"][i])); feats["object_id"].append(lab)
                feats["flow"].append(tuple(r["flow"][i])); feats["predicted_kp"].append(tuple(r["predicted_kp"][i]))
            idx = per_obj[lab]
            tracked[lab] = dict(age=np.array([int(r["new_age"][i]) for i in idx]), kp=np.array([prev_dynamic["predicted_kp"][i] for i in idx], np.float64))
    to_sample, why = requires_sampling(boundary["objects"], boundary["inner_boxes"], tracked, set(status), max_age, age_buffer, min_tracks, min_iou)
    for o in to_sample:
        s = status.setdefault(o, _status())
        s["object_resampled"] = True
        if why[o]["new"]:
            s["object_new"] = True
    if to_sample:
        need = [max(max_features - status[o]["num_track"], 0) for o in to_sample]
        sm = sample_dynamic(motion_mask, flow, det, to_sample, need, shrink_row, shrink_col, 0.01, tid)
        tid = sm["next_tracklet_id"]
        for o, nz, ns, nc in zip(to_sample, sm["n_zero_flow"], sm["n_sampled"], sm["n_candidates"]):
            status[o]["num_zero_flow"] += int(nz)
            if nc > 0:
                status[o]["num_sampled"] = int(ns)
        for i in range(len(sm["label"])):
            feats["tracklet_id"].append(int(sm["tracklet_id"][i])); feats["kp"].append(tuple(sm["kp"][i])); feats["age"].append(0)
            feats["object_id"].append(int(sm["label"][i])); feats["flow"].append(tuple(sm["flow"][i])); feats["predicted_kp"].append(tuple(sm["predicted_kp"][i]))
    out = dict(tracklet_id=np.array(feats["tracklet_id"], np.int64), kp=np.array(feats["kp"], np.float64).reshape(-1, 2), age=np.array(feats["age"], np.int64),
               object_id=np.array(feats["object_id"], np.int32), flow=np.array(feats["flow"], np.float64).reshape(-1, 2),
               predicted_kp=np.array(feats["predicted_kp"], np.float64).reshape(-1, 2))
    return out, to_sample, status, tid


def _disc(mask, x, y, r, value):
    """cv::circle(mask, (x, y), r, value, FILLED): rows of half-width floor(sqrt(r^2 + r - dy^2)) [recalled, as flow_oracle.track_dynamic]"""
    h, w = mask.shape
    for dy in range(-r, r + 1):
        yy, v = y + dy, r * r + r - dy * dy
        if yy < 0 or yy >= h or v < 0:
            continue
        hw = int(np.floor(np.sqrt(v)))
        mask[yy, max(0, x - hw):min(w - 1, x + hw) + 1] = value


def track_dynamic_klt_frame(prev_dynamic, prev_gray, gray, motion_mask, boundary, next_tracklet_id, max_features=50, max_age=25, age_buffer=3,
                            min_tracks=20, min_iou=0.3, min_distance=2, shrink_row=0, shrink_col=0):
    """FeatureTracker::trackDynamicKLT (FeatureTracker.cc:500-862), the dynamic tracker used when no dense flow is provided
    (params_.prefer_provided_optical_flow false, :125-140), for one frame:
      :595-706  forward pyramidal LK (21x21, 3 levels, 30 iterations; klt_oracle.calc_pyr_lk - the reference runs
                cv::cuda::SparsePyrLKOpticalFlow, whose arithmetic is not restated: the CPU lkpyramid restatement stands in) of the
                previous frame's usable dynamic features k-1 -> k; per tracked point (status set): label under the point, detection
                mask test (before any count), info_ bookkeeping, contained / object / same-label / shrunken-image tests, age + 1
                (dropped beyond max_dynamic_feature_age), discs into the detection mask
      :766-772  requiresSampling (same function as the dense-flow tracker)
      :774-861  per object to sample: goodFeaturesToTrack(mono, 50, 0.01, min_distance, (mask == object) & detection mask),
                ANMS RangeTree to max_features - num_track (tolerance 0.01), new features of age 0 inside the shrunken image
    prev_dynamic: None or dict(tracklet_id, kp, age, object_id) of frame k-1; gray images uint8.
      dynosam/src/frontend/vision/StaticFeatureTracker.cc
      :244-300   trackStatic       first frame -> detectFeatures, else trackPoints           (track_static_frame below)
      :420-637   trackPoints       LK + flow-back (klt_oracle), RANSAC homography (ransac_oracle), usable / age tests,
                                   outliers, top-up below min_features_per_frame
      :330-418   detectFeatures    detection mask (caller mask & background & discs), SparseFeatureDetector::detect
                                   (FeatureDetector.cc:186-241: CLAHE -> corners -> ANMS -> cornerSubPix: clahe_oracle, gftt_oracle,
                                   anms_range_tree above, subpix_oracle), contained / shrunken / background tests, new ids
Undefined in the reference and fixed here (and in the product): objects are sampled in ascending id (the reference fills a
    tbb::concurrent_unordered_map), corners keep the detector's order through the response sort (all responses are equal), a tracked
    point outside the image is dropped (the reference indexes the mask out of bounds).  The features of frame k carry no flow in this
    mode (the reference writes kp_k - kp_{k-1} into the PREVIOUS frame's feature).  PARITY UNPINNED against the binary."""
    from . import gftt_oracle as GO
    from . import klt_oracle as KO
    h, w = motion_mask.shape
    status, tracked = {}, {}
    det = np.asarray(boundary["boundary_mask"], np.uint8).copy()
    feats = dict(tracklet_id=[], kp=[], age=[], object_id=[])
    tid = int(next_tracklet_id)
    if prev_dynamic is not None and len(prev_dynamic["tracklet_id"]):
        prev_pts = np.asarray(prev_dynamic["kp"], np.float64).reshape(-1, 2).astype(np.float32)     # toOpenCV: cv::Point2f
        cur, st = KO.calc_pyr_lk(prev_gray, gray, prev_pts, None, 3, 30, 0.03)
        per_obj = {}
        for i in range(len(prev_pts)):
            if not st[i]:
                continue
            kx, ky = float(cur[i][0]), float(cur[i][1])
            x, y = int(kx), int(ky)
            if not (0 <= x < w and 0 <= y < h):
                continue
            lab = int(motion_mask[y, x])
            if det[y, x] == 0:
                continue
            prev_lab = int(prev_dynamic["object_id"][i])
            s = status.setdefault(lab, _status())
            s["num_previous_track"] += 1
            if lab == 0:
                s["num_tracked_with_background_label"] += 1
            if lab != prev_lab:
                s["num_tracked_with_different_label"] += 1
            contained = kx >= 0.0 and kx < w and ky >= 0.0 and ky < h
            if not (contained and lab != 0 and lab == prev_lab):
                continue
            if not bool(within_shrunken(x, y, w, h, shrink_row, shrink_col)):
                s["num_outside_shrunken_image"] += 1
                continue
            age = int(prev_dynamic["age"][i]) + 1
            if age > max_age:
                continue
            per_obj.setdefault(lab, []).append((int(prev_dynamic["tracklet_id"][i]), (kx, ky), age))
            s["num_track"] += 1
            _disc(det, x, y, min_distance, 0)
        for lab in sorted(per_obj):
            for t_, kp_, a_ in per_obj[lab]:
                feats["tracklet_id"].append(t_); feats["kp"].append(kp_); feats["age"].append(a_); feats["object_id"].append(lab)
            tracked[lab] = dict(age=np.array([a for _, _, a in per_obj[lab]]), kp=np.array([k for _, k, _ in per_obj[lab]], np.float64))
    to_sample, why = requires_sampling(boundary["objects"], boundary["inner_boxes"], tracked, set(status), max_age, age_buffer, min_tracks, min_iou)
    for o in to_sample:
        s = status.setdefault(o, _status())
        s["object_resampled"] = True
        if why[o]["new"]:
            s["object_new"] = True
    for o in to_sample:
        combined = ((motion_mask == o) & (det != 0)).astype(np.uint8) * 255
        corners, _ = GO.good_features_to_track(gray, combined, max_features, 0.01, float(min_distance))
        need = max(max_features - status[o]["num_track"], 0)
        if len(corners) == 0:
            continue                                  # suppressNonMax returns before anything is counted; num_sampled = 0 then
        keep = anms_range_tree(corners, need, 0.01, w, h)
        status[o]["num_sampled"] = len(keep)
        for i in keep:
            kx, ky = float(corners[i][0]), float(corners[i][1])
            if not bool(within_shrunken(int(kx), int(ky), w, h, shrink_row, shrink_col)):
                continue
            feats["tracklet_id"].append(tid); tid += 1
            feats["kp"].append((kx, ky)); feats["age"].append(0); feats["object_id"].append(int(o))
    out = dict(tracklet_id=np.array(feats["tracklet_id"], np.int64), kp=np.array(feats["kp"], np.float64).reshape(-1, 2), age=np.array(feats["age"], np.int64),
               object_id=np.array(feats["object_id"], np.int32))
    return out, to_sample, status, tid


# ---- the static half of FeatureTracker::track: KltFeatureTracker::trackStatic composed from the per-stage oracles ----
def _usable_static(kp, motion_mask, shrink_row=0, shrink_col=0):
    h, w = motion_mask.shape
    ok = (kp[:, 0] >= 0) & (kp[:, 0] < w) & (kp[:, 1] >= 0) & (kp[:, 1] < h)
    ok &= within_shrunken(kp[:, 0], kp[:, 1], w, h, shrink_row, shrink_col)         # FeatureTrackerBase.cc:313-326: truncated coordinates, STRICT inequalities
    x, y = np.floor(kp[:, 0]).astype(int), np.floor(kp[:, 1]).astype(int)
    ok[ok] &= motion_mask[y[ok], x[ok]] == 0
    return ok


def detect_static_features(gray, motion_mask, current, detection_mask, next_tracklet_id, max_features=400, max_before_anms=2000, quality_level=0.001,
                           min_distance=8, shrink_row=0, shrink_col=0, use_anms=True, use_clahe=True, use_subpix=True, detector=0, orb=None, gfft=(3, False, 0.04),
                           anms=(4, 5, 5, None), subpix=(5, 5, -1, -1)):
    """KltFeatureTracker::detectFeatures.  subpix = (window half width, half height, zero zone width, height).  anms = (AnmsAlgorithmType, nr_horizontal_bins, nr_vertical_bins, binning_mask). current: dict(tracklet_id, kp [n,2] f64, age). returns (dict, next id).
    detector: TrackerParams::FeatureDetectorType (0 GFTT, 1 ORB_SLAM_ORB with orb = orb_oracle.OrbParams or None = the defaults)"""
    from . import clahe_oracle as CO, gftt_oracle as GO, orb_oracle as OO, subpix_oracle as SO
    mask = np.full(motion_mask.shape, 255, np.uint8) if detection_mask is None else np.array(detection_mask, np.uint8)
    mask[motion_mask != 0] = 0
    for x, y in current["kp"]:
        # cv::circle(..., cv::Point2f(kp(0), kp(1)), ...): Point2f -> Point = saturate_cast<int> (to nearest, ties to even)
        _disc(mask, int(np.rint(np.float32(x))), int(np.rint(np.float32(y))), min_distance, 0)
    want = max_features - len(current["tracklet_id"])
    if want <= 0:
        return current, next_tracklet_id
    img = CO.clahe(gray) if use_clahe else gray
    if detector == 1:
        # FeatureDetector.cc:124-145: keypoints only, no mask; NonMaximumSupression.cc:45-57: by (int)response, descending, in front of ANMS
        c, resp, _, _, _ = OO.detect(img, orb or OO.OrbParams(nfeatures=max_before_anms), with_angle=False)
    else:
        c, _ = GO.good_features_to_track(img, mask, max_before_anms, quality_level, float(min_distance), *gfft)
        resp = None                                                     # cv::GFTTDetector leaves KeyPoint::response at 0
    if use_anms:
        c = c[suppress_non_max(c, resp, want, 0.1, motion_mask.shape[1], motion_mask.shape[0], *anms)]
    if use_subpix and len(c):
        c, _ = SO.corner_sub_pix(img, c, subpix[0], win_h=subpix[1], zero_zone=(subpix[2], subpix[3]))
    c = c.astype(np.float64).reshape(-1, 2)
    c = c[_usable_static(c, motion_mask, shrink_row, shrink_col)]      # (without ANMS every raw keypoint: FeatureDetector.cc:201-222)
    ids = next_tracklet_id + np.arange(len(c), dtype=np.int64)
    out = dict(tracklet_id=np.concatenate([current["tracklet_id"], ids]), kp=np.concatenate([current["kp"].reshape(-1, 2), c]),
               age=np.concatenate([current["age"], np.zeros(len(c), np.int64)]))
    return out, next_tracklet_id + len(c)


def track_static_frame(previous, prev_gray, gray, motion_mask, detection_mask, next_tracklet_id, max_features=400, min_features=200, max_age=25,
                       max_before_anms=2000, quality_level=0.001, min_distance=8, shrink_row=0, shrink_col=0, use_anms=True, use_clahe=True,
                       use_subpix=True, geometric_verification=True, ransac_threshold=5.0, R_km1_k=None, K=None, detector=0, orb=None, gfft=(3, False, 0.04),
                       anms=(4, 5, 5, None), subpix=(5, 5, -1, -1)):
    """KltFeatureTracker::trackStatic. previous: None or dict(tracklet_id, kp, age). returns (features dict, outlier ids, info dict, next id)"""
    from . import klt_oracle as KO, ransac_oracle as RO
    kw = dict(max_features=max_features, max_before_anms=max_before_anms, quality_level=quality_level, min_distance=min_distance, shrink_row=shrink_row,
              shrink_col=shrink_col, use_anms=use_anms, use_clahe=use_clahe, use_subpix=use_subpix, detector=detector, orb=orb, gfft=gfft, anms=anms, subpix=subpix)
    info = dict(static_track_optical_flow=0, static_track_detections=0, new_static_detections=False, static_track_ransac_rejected=0)
    empty = dict(tracklet_id=np.zeros(0, np.int64), kp=np.zeros((0, 2)), age=np.zeros(0, np.int64))
    if previous is None or len(previous["tracklet_id"]) == 0:
        out, nid = detect_static_features(gray, motion_mask, empty, detection_mask, next_tracklet_id, **kw)
        info["static_track_detections"] = len(out["tracklet_id"])
        return out, np.zeros(0, np.int64), info, nid
    prev_kp = previous["kp"].astype(np.float32)
    # the predicted rotation of FeatureTracker::track: LK from predictKeypointsGivenRotation, retried cold below 10 successes (StaticFeatureTracker.cc:455-503)
    init = None if R_km1_k is None else KO.predict_keypoints_given_rotation(prev_kp, R_km1_k, K, gray.shape[1], gray.shape[0], shrink_row, shrink_col)
    cur, _back, good, _st = KO.track_points(prev_gray, gray, prev_kp, init)
    good = good.astype(bool)
    if geometric_verification and good.any():
        gi = np.nonzero(good)[0]
        inl, _best, _H = RO.verify_homography(prev_kp[gi], cur[gi], ransac_threshold)
        good[gi[inl == 0]] = False
        info["static_track_ransac_rejected"] = int((inl == 0).sum())
    outliers = determine_outlier_ids(previous["tracklet_id"][good], previous["tracklet_id"])       # StaticFeatureTracker.cc:600-606
    kp = cur.astype(np.float64)
    keep = good & _usable_static(kp, motion_mask, shrink_row, shrink_col) & (previous["age"] + 1 <= max_age)
    tracked = dict(tracklet_id=previous["tracklet_id"][keep], kp=kp[keep], age=previous["age"][keep] + 1)
    info["static_track_optical_flow"] = int(keep.sum())
    nid = next_tracklet_id
    if len(tracked["tracklet_id"]) < min_features:
        n0 = len(tracked["tracklet_id"])
        tracked, nid = detect_static_features(gray, motion_mask, tracked, detection_mask, nid, **kw)
        info["new_static_detections"] = True
        info["static_track_detections"] = len(tracked["tracklet_id"]) - n0
    return tracked, outliers, info, nid
