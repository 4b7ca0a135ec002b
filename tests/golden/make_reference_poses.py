#!/usr/bin/env python3
"""A few frames of the reference's own pose files (BASELINE configs 2-5 name them), as data:

    load/peoplesnapshot/male-3-casual/poses/anim_nerf_train.npz   frames 0, 40, 80, 113 (114 training frames)
    load/animation/aist/poses.npz                                 frames 0, 100, 200, 319 (320 out-of-distribution frames),
                                                                  translation re-based as datasets/animation.py:129-130 does

  python tests/golden/make_reference_poses.py        (needs /root/reference) -> tests/golden/reference_poses.npz

bench.py, tools/relight_bench.py and the tests drive them through plain forward kinematics (synthetic.make_rig / smpl.py)
into tfs / w2s -- SURVEY 8(d) "Deformer"."""
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def main():
    a = np.load(f"{REF}/load/peoplesnapshot/male-3-casual/poses/anim_nerf_train.npz")
    fa = np.array([0, 40, 80, 113])
    b = np.load(f"{REF}/load/animation/aist/poses.npz")
    fb = np.array([0, 100, 200, 319])
    transl = b["trans"] - b["trans"][0:1] + np.array([0.0, 0.15, 5.0])          # datasets/animation.py:129-130
    out = {"male-3-casual_frames": fa, "male-3-casual_global_orient": a["global_orient"][fa].astype(np.float32),
           "male-3-casual_body_pose": a["body_pose"][fa].astype(np.float32), "male-3-casual_transl": a["transl"][fa].astype(np.float32),
           "aist_frames": fb, "aist_global_orient": b["poses"][fb, :3].astype(np.float32), "aist_body_pose": b["poses"][fb, 3:72].astype(np.float32),
           "aist_transl": transl[fb].astype(np.float32)}
    np.savez_compressed(f"{HERE}/reference_poses.npz", **out)
    print("reference_poses.npz", os.path.getsize(f"{HERE}/reference_poses.npz"), "bytes")


if __name__ == "__main__":
    main()
