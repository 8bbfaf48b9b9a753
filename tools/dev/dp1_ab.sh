# one-rank RCCL step: no collectives / every sum through the backend's stream / gradient slices in-stream, then the two small sums too
export VDN_DP_ASYNC_GROUP=0
for r in 1 2 3; do
DP1_OFF=1 python3 tools/dev/dp1_wall.py off 2>&1 | grep "one-rank\|exposed"
VDN_DP_INSTREAM=1 python3 tools/dev/dp1_wall.py instream 2>&1 | grep "one-rank\|exposed"
VDN_DP_INSTREAM=1 VDN_DP_EIK_INSTREAM=1 python3 tools/dev/dp1_wall.py instream_eik 2>&1 | grep "one-rank\|exposed"
VDN_DP_INSTREAM=1 VDN_DP_EIK_INSTREAM=1 VDN_DP_COUNT_INSTREAM=1 python3 tools/dev/dp1_wall.py instream_all 2>&1 | grep "one-rank\|exposed"
VDN_DP_INSTREAM=1 VDN_DP_EIK_INSTREAM=1 VDN_DP_SIDE_GROUP=0 python3 tools/dev/dp1_wall.py instream_eik_1comm 2>&1 | grep "one-rank\|exposed"
done
