"""TEST INFRASTRUCTURE - not part of the product (only tests/ may import this).

A restatement of `cv2.decomposeProjectionMatrix`, the one third-party routine on the reference's camera path
(`/root/reference/dpt_models/dataset.py:25`, called from `load_K_Rt_from_P`, dataset.py:14-35, for every camera of a scene:
dataset.py:84-91 and poses.py:142-149). OpenCV is absent from this image (`import cv2` fails) and not vendored by the
reference, which pins `opencv_python==4.5.2.52` (README.md:10). What follows is OpenCV's published algorithm for that version
(modules/calib3d/src/calibration.cpp: `cvDecomposeProjectionMatrix` and `cvRQDecomp3x3`), step for step:

  * the camera position is the null vector of P: the last row of V^T of the SVD of P padded to 4x4 with a zero row
    (returned homogeneous, 4x1; the reference divides by its last entry, dataset.py:35);
  * the left 3x3 block M is RQ-decomposed, M = R Q, by three Givens rotations applied from the right - Qx zeroes M[2,1],
    Qy zeroes M[2,0], Qz zeroes M[1,0], each normalised with `1 / sqrt(c^2 + s^2 + DBL_EPSILON)` - followed by OpenCV's
    resolution of the sign ambiguity: the first two diagonal entries of R are made positive by a further 180-degree
    rotation about z, y or x (the LAST diagonal entry keeps whatever sign det(M) leaves it);
  * all of it in double precision whatever the input type; the outputs are converted back to the input's type
    (a float32 P - what the reference passes, dataset.py:84-88 - returns float32 K, R and position).

PARITY UNPINNED against OpenCV itself: the library cannot be run here and the reference holds no golden vectors for this
routine. The restatement is pinned by its own properties (tests/test_dataset_cpu.py: M = K R to 1e-12, R a proper rotation,
K upper-triangular, P @ position = 0) and is what `vdn_train.dataset.load_K_Rt_from_P` (scipy's Householder RQ + an explicit
sign convention) is compared with, through the reference's own post-processing (dataset.py:27-35, restated in
`load_K_Rt_from_P_reference` below)."""
import numpy as np

DBL_EPSILON = float(np.finfo(np.float64).eps)


def rq_decomp_3x3(M):
    """cvRQDecomp3x3: M[3,3] -> (R upper-triangular, Q orthogonal) with M = R Q (float64)."""
    M = np.array(M, dtype=np.float64)

    def givens(c, s):
        z = 1.0 / np.sqrt(c * c + s * s + DBL_EPSILON)
        return c * z, s * z

    # x axis: Qx = [[1,0,0],[0,c,s],[0,-s,c]], c = m33 / |(m32, m33)|, s = m32 / |(m32, m33)|
    c, s = givens(M[2, 2], M[2, 1])
    Qx = np.array([[1, 0, 0], [0, c, s], [0, -s, c]], dtype=np.float64)
    R = M @ Qx
    R[2, 1] = 0.0
    # y axis: Qy = [[c,0,-s],[0,1,0],[s,0,c]], c = r33 / |(r31, r33)|, s = -r31 / |(r31, r33)|
    c, s = givens(R[2, 2], -R[2, 0])
    Qy = np.array([[c, 0, -s], [0, 1, 0], [s, 0, c]], dtype=np.float64)
    M = R @ Qy
    M[2, 0] = 0.0
    # z axis: Qz = [[c,s,0],[-s,c,0],[0,0,1]], c = m22 / |(m21, m22)|, s = m21 / |(m21, m22)|
    c, s = givens(M[1, 1], M[1, 0])
    Qz = np.array([[c, s, 0], [-s, c, 0], [0, 0, 1]], dtype=np.float64)
    R = M @ Qz
    R[1, 0] = 0.0
    # the decomposition's ambiguity: the diagonal entries of R except the last one shall be positive
    if R[0, 0] < 0:
        if R[1, 1] < 0:           # 180 degrees about z: diag(-1, -1, 1)
            R[0, 0] *= -1; R[0, 1] *= -1; R[1, 1] *= -1
            Qz[0, 0] *= -1; Qz[0, 1] *= -1; Qz[1, 0] *= -1; Qz[1, 1] *= -1
        else:                     # 180 degrees about y: diag(-1, 1, -1)
            R[0, 0] *= -1; R[0, 2] *= -1; R[1, 2] *= -1; R[2, 2] *= -1
            Qz = Qz.T.copy()
            Qy[0, 0] *= -1; Qy[0, 2] *= -1; Qy[2, 0] *= -1; Qy[2, 2] *= -1
    elif R[1, 1] < 0:             # 180 degrees about x: diag(1, -1, -1)
        R[0, 1] *= -1; R[0, 2] *= -1; R[1, 1] *= -1; R[1, 2] *= -1; R[2, 2] *= -1
        Qz = Qz.T.copy()
        Qy = Qy.T.copy()
        Qx[1, 1] *= -1; Qx[1, 2] *= -1; Qx[2, 1] *= -1; Qx[2, 2] *= -1
    Q = Qz.T @ Qy.T @ Qx.T
    return R, Q


def decompose_projection_matrix(P):
    """cv2.decomposeProjectionMatrix(P)[:3]: P[3,4] -> (cameraMatrix [3,3], rotMatrix [3,3], transVect [4,1]) in P's dtype."""
    P = np.asarray(P)
    out_dtype = P.dtype if P.dtype in (np.float32, np.float64) else np.float64
    P64 = np.asarray(P, dtype=np.float64)
    if P64.shape != (3, 4):
        raise ValueError("projection matrix must be 3x4")
    padded = np.zeros((4, 4), dtype=np.float64)
    padded[:3] = P64
    _, _, Vt = np.linalg.svd(padded)
    position = Vt[3].reshape(4, 1)
    K, R = rq_decomp_3x3(P64[:, :3])
    return K.astype(out_dtype), R.astype(out_dtype), position.astype(out_dtype)


def load_K_Rt_from_P_reference(P):
    """dataset.py:25-35 on the routine above: -> (intrinsics [4,4] float64, pose [4,4] float32 camera-to-world)."""
    K, R, t = decompose_projection_matrix(P)
    K = K / K[2, 2]
    intrinsics = np.eye(4)
    intrinsics[:3, :3] = K
    pose = np.eye(4, dtype=np.float32)
    pose[:3, :3] = R.transpose()
    pose[:3, 3] = (t[:3] / t[3])[:, 0]
    return intrinsics, pose
