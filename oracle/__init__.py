"""CPU oracle for the `app/ml` hot path of malak29/video-text-detection-system.

TEST INFRASTRUCTURE ONLY.  Nothing under ``video-text-detection-system_amd/`` imports this package;
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg do, and there
only as the checker / the timed baseline -- never as the thing shipped.

Pinning status:
  * preprocess (Pillow antialiased bilinear)  -- pinned bit-exact against PIL (tests/test_oracle_preprocess.py)
  * CRNN, DBHead, FPN sub-convs, vocab, CTC decode, pipeline harness rows H1-H3
                                              -- pinned against the reference's own classes loaded from
                                                 /root/reference in the build container; vectors committed
                                                 under tests/golden/ with tests/golden/make_golden.py
  * ResNet trunk (torchvision 0.16.1, absent) -- PARITY UNPINNED (restated from the public ResNet v1.5 definition)
  * OpenCV findContours / contourArea / minAreaRect / boxPoints / resize (cv2 absent)
                                              -- PARITY UNPINNED (restated from the published algorithms)
"""
