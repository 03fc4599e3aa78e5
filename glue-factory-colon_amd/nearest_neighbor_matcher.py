"""Nearest-neighbour matcher on MI355X -- drop-in for `gluefactory.models.matchers.nearest_neighbor_matcher`
(reference gluefactory/models/matchers/nearest_neighbor_matcher.py:47-79): same configuration keys
(`ratio_thresh`, `distance_thresh`, `mutual_check`), same prediction dictionary.  The training loss
(`N_pair`) is out of scope.

    model.matcher.name = glue_factory_colon_amd.nearest_neighbor_matcher
"""
import torch

from . import _native as nat
from .base_model import BaseModel, conf_get


class NearestNeighborMatcher(BaseModel):
    default_conf = {"ratio_thresh": None, "distance_thresh": None, "mutual_check": True, "loss": None}
    required_data_keys = ["descriptors0", "descriptors1"]

    def _init(self, conf):
        if conf_get(conf, "loss") is not None:
            raise NotImplementedError("training losses are out of scope (inference path)")
        self._ws = nat.Workspace()
        self.set_initialized()

    def _forward(self, data):
        d0, d1 = data["descriptors0"], data["descriptors1"]
        nat.require_cuda(d0, "data['descriptors0']")
        d0, d1 = d0.float().contiguous(), d1.float().contiguous()
        b, m, d = d0.shape
        n = d1.shape[1]
        dev = d0.device
        m0 = torch.full((b, m), -1, device=dev, dtype=torch.long)
        m1 = torch.full((b, n), -1, device=dev, dtype=torch.long)
        ms0, ms1 = torch.zeros((b, m), device=dev), torch.zeros((b, n), device=dev)
        sim = torch.zeros((b, m, n), device=dev)
        la = torch.zeros((b, m + 1, n + 1), device=dev)
        if m > 0 and n > 0:
            lib = nat.lib()
            ws = self._ws.get(lib.gfc_nn_workspace_bytes(b, m, n), dev)
            nat.check(lib.gfc_nn_match(nat.ptr(d0), nat.ptr(d1), b, m, n, d, float(conf_get(self.conf, "ratio_thresh") or 0),
                                       float(conf_get(self.conf, "distance_thresh") or 0),
                                       int(bool(conf_get(self.conf, "mutual_check"))), nat.ptr(m0), nat.ptr(m1),
                                       nat.ptr(ms0), nat.ptr(ms1), nat.ptr(sim), nat.ptr(la), nat.ptr(ws), ws.numel(),
                                       nat.stream_ptr(dev)), "gfc_nn_match")
        return {"matches0": m0, "matches1": m1, "matching_scores0": ms0, "matching_scores1": ms1, "similarity": sim,
                "log_assignment": la}

    def loss(self, pred, data):
        raise NotImplementedError


__main_model__ = NearestNeighborMatcher
