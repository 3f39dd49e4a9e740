"""A randomly labelled path graph: exercises `model.assign_random_labels` (model.py:475-539 of the reference) with a small
vulnerability library.  `build(m, ...)` takes the model module to build with, like kitchen_sink.build, so that the fixture
generator can run it against the reference's classes and the tests against this package's.  A directed path keeps every
node at one successor at most (see assign_random_labels on successor order)."""
from __future__ import annotations

import random


class _Path:
    """0 -> 1 -> ... -> n-1, with the two accessors assign_random_labels needs (a networkx.DiGraph has them too)."""

    def __init__(self, n: int):
        self.nodes = list(range(n))
        self._edges = [(i, i + 1) for i in range(n - 1)]

    def edges(self):
        return list(self._edges)


def library(m, n: int):
    L, R = m.VulnerabilityType.LOCAL, m.VulnerabilityType.REMOTE
    return {
        "UacBypass": m.VulnerabilityInfo(description="", type=L, outcome=m.AdminEscalation(), precondition=m.Precondition("Windows&Win10"), cost=2.0),
        "Fingerprint": m.VulnerabilityInfo(description="", type=R, outcome=m.CustomerData(), cost=1.0),
        "BruteRdp": m.VulnerabilityInfo(description="", type=R, outcome=m.LateralMove(), precondition=m.Precondition("Windows&PortRDPOpen"), cost=9.0),
        "ReadNotes": m.VulnerabilityInfo(description="", type=L, outcome=m.LeakedCredentials([m.CachedCredential(str(n - 1), "RDP", "pw")]), cost=3.0),
    }


def build(m, seed: int, n: int = 6, graph=None):
    lib = library(m, n)
    if graph is None:
        graph = _Path(n)
    random.seed(seed)
    network = m.assign_random_labels(graph, lib, m.SAMPLE_IDENTIFIERS)
    identifiers = m.Identifiers(properties=list(m.SAMPLE_IDENTIFIERS.properties), ports=list(m.SAMPLE_IDENTIFIERS.ports),
                                local_vulnerabilities=["UacBypass", "ReadNotes", "RecentlyAccessedMachines"],
                                remote_vulnerabilities=["Fingerprint", "BruteRdp"])
    return m.Environment(network=network, vulnerability_library={}, identifiers=identifiers)
