"""TEST STUB of the Detectron2 package (tests/test_d2_seam.py puts tests/stubs on sys.path).

Detectron2 is not installable here (no network), so the seam between locov_amd and the objects a real
train_ovnet.py run hands it -- detectron2.structures.{Boxes, Instances}, detectron2.modeling.roi_heads.ROI_HEADS_REGISTRY,
detectron2.modeling.postprocessing.detector_postprocess, detectron2.utils.events.get_event_storage -- is exercised against
this minimal restatement of Detectron2's PUBLIC API ([D2-upstream]; only what the LSM ROI-head path touches).  The classes
are deliberately DISTINCT from locov_amd.structures' so that any place that builds the wrong type is caught."""
__version__ = "0.6-stub"
