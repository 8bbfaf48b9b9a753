"""The learnable-camera modules (reference dpt_models/poses.py:16-93, lie_group_helper.py:47-83) on the CPU: they are plain
torch parameter holders; known answers from the reference's own make_c2w (tests/golden/raygrad.npz: pose/*)."""
import numpy as np
import torch

from conftest import load_golden


def test_make_c2w_matches_the_reference_vectors():
    from dpt_models.lie_group_helper import make_c2w
    fx = load_golden("raygrad")
    for r, t, want in zip(fx["pose/r"], fx["pose/t"], fx["pose/c2w"]):
        got = make_c2w(torch.tensor(r), torch.tensor(t)).numpy()
        assert got.shape == (4, 4) and np.array_equal(got, want)


def test_learnpose_is_a_delta_on_the_initial_pose_and_differentiable():
    from dpt_models.poses import LearnPose, LearnIntrin
    init = torch.eye(4).repeat(3, 1, 1)
    init[:, :3, 3] = torch.tensor([[1.0, 2.0, 3.0], [0.0, 0.0, 1.0], [4.0, 5.0, 6.0]])
    net = LearnPose(3, True, False, init_c2w=init)
    assert [n for n, p in net.named_parameters() if p.requires_grad] == ["r"]
    assert set(net.state_dict()) == {"init_c2w", "r", "t"}                  # poses.py:31-35
    assert torch.equal(net(1), init[1])                                     # zero delta: the regularised Exp(0) is exactly I
    with torch.no_grad():
        net.r[2] = torch.tensor([0.0, 0.0, np.pi / 2])
    c2w = net(2)
    assert torch.allclose(c2w[:3, :3], torch.tensor([[0.0, -1.0, 0.0], [1.0, 0.0, 0.0], [0.0, 0.0, 1.0]]), atol=1e-6)
    assert torch.allclose(c2w[:3, 3], torch.tensor([-5.0, 4.0, 6.0]), atol=1e-5)   # R applied to the initial translation
    c2w[:3, 3].sum().backward()
    assert net.r.grad[2].abs().sum() > 0 and net.r.grad[0].abs().sum() == 0
    k = LearnIntrin(600, 800, req_grad=True, order=2, init_focal=torch.tensor(1000.0))()
    assert k.shape == (4, 4) and abs(float(k[0, 0]) - 1000.0) < 1e-3 and float(k[0, 2]) == 400.0 and float(k[1, 2]) == 300.0
    assert not k.requires_grad                                              # poses.py:80-93: built from fx.item()


def test_the_reference_shipped_pose_checkpoints_load_and_give_the_reference_cameras():
    """The reference ships learned cameras: pretrained-models/*/*/pnf_300000.pth (dpt_runner.py:383-401; ten of them, 28 - 40
    cameras each). tests/golden/pnf_rays.npz (make_golden.py::pnf_fixture) holds their parameters and what the REFERENCE's own
    LearnPose / LearnIntrin return for them: this repo's modules take the same state_dicts (strict) and return the same
    camera-to-world matrices - bit for bit on the CPU - and the same intrinsics."""
    from dpt_models.poses import LearnPose, LearnIntrin
    fx = load_golden("pnf_rays")
    H, W = int(fx["H"]), int(fx["W"])
    assert len(fx["names"]) == 10
    cams = 0
    for tag in fx["names"]:
        tag = str(tag)
        n = fx[tag + "/r"].shape[0]
        pose = LearnPose(n, True, True, torch.zeros(n, 4, 4))
        pose.load_state_dict({k: torch.tensor(fx["%s/%s" % (tag, k)]) for k in ("init_c2w", "r", "t")}, strict=True)
        intr = LearnIntrin(H, W, req_grad=True)
        intr.load_state_dict({"fx": torch.tensor(fx[tag + "/fx"])}, strict=True)
        with torch.no_grad():
            got = np.stack([pose(i).numpy() for i in range(n)])
        assert np.array_equal(got, fx[tag + "/c2w"]), tag
        assert np.array_equal(intr().numpy(), fx[tag + "/intrinsic"]), tag
        # the learned cameras are proper rigid motions around the scene's unit sphere
        R = got[:, :3, :3]
        assert np.abs(np.einsum("nij,nkj->nik", R, R) - np.eye(3)).max() < 1e-5
        assert 1.5 < np.linalg.norm(got[:, :3, 3], axis=1).min() and np.linalg.norm(got[:, :3, 3], axis=1).max() < 3.5
        cams += n
    assert cams == 340
