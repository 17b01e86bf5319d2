"""CPU oracle for the LiDAR4D ray-rendering hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``lidar4d_amd/`` may import this package; only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg do, and there
only as the checker / the timed CPU baseline -- never as the thing shipped.

Contents
--------
``tcnn_ref``    restatement of the tiny-cuda-nn operators LiDAR4D calls (HashGrid and Frequency
                encodings, FullyFusedMLP).  tiny-cuda-nn is an un-vendored, UNPINNED dependency of
                the reference (``git clone`` of master, reference README.md:88-91) and is absent
                from /root/reference, so this part of the oracle is **parity unpinned**: it follows
                the published algorithm and is anchored on the reference's
                call sites (model/hash_field.py:47-57,107-117; model/flow_field.py:67-77;
                model/lidar4d.py:68-117), on self-consistency tests (dense-vs-hashed agreement,
                partition of unity, fp64 recomputation, finite differences) and nothing stronger.
``fields_ref``  restatement of the reference's own torch code on the path: renderer.py,
                hash_field.py, planes_field.py, flow_field.py, activation.py, lidar4d.py.  This
                part IS pinned: ``oracle/make_golden.py`` imports the real reference modules from a
                scratch copy of /root/reference (in the build container only) and the fixtures it
                writes to ``tests/golden`` are checked against this restatement by
                ``tests/test_oracle_golden.py``.
``rays_ref``    restatement of data/base_dataset.py:get_lidar_rays + the synthetic
                KITTI-360-shaped frame of SURVEY.md section 8(d).
"""
