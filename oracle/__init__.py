"""oracle/ -- CPU restatement of the SegHiero training hot path.  TEST INFRASTRUCTURE ONLY.

This package is the *checker*, never the product: only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may import it.
``seghiero_amd`` (the product) never imports anything from here and fails loudly when its
HIP library is missing.

What it is: plain ``torch`` (CPU, fp32; f64 where the reference uses f64) modules and
functions written from the formulas of SURVEY.md Appendix A, mirroring the reference's class
names, constructor/forward signatures and state_dict keys:

* ``oracle.hierarchy``  -- YAML range lists -> index tensors   (reference ``train.py:52-99``)
* ``oracle.nets``       -- ResNetBackbone (torchvision-free), DepthwiseSeparableASPPContrastHead
                           (reference ``models/backbone/resnet.py:26-75``,
                           ``models/head/sep_aspp_contrast_head.py:6-254``)
* ``oracle.losses``     -- HieraTripletLoss, TreeTripletLoss, CrossEntropyLoss,
                           RMIHieraTripletLoss (reference ``models/loss/*.py``)
* ``oracle.step``       -- one training step (reference ``train.py:260-320``) and the
                           validation pixel-accuracy / build-defined mIoU metric.

Pinning status (see DESIGN.md "Oracle"): the head and every loss are pinned against golden
vectors produced in the authoring container by importing the reference itself
(``tools/make_goldens.py`` -> ``tests/golden/*.npz``; ``tests/test_oracle_golden.py``).
The ResNet trunk comes from torchvision, which is absent from ``/root/reference`` and from this
image: that part is restated from the public architecture and is **parity unpinned** beyond
parameter counts / state_dict key shapes.
"""
