"""Random traffic-graph topologies (`CyberBattleRandom-v0`).

Restated from src/CyberBattleSim/cyberbattle/simulation/generate_network.py: identifiers :15-20, the traffic graph
(:23-76: per protocol a two-block stochastic block model whose edge probabilities come from a beta distribution), the
CyberBattle model planted on it (:79-263) and `new_environment` (:266-294).  Result parity with the reference for a given
seed needs the same pseudo-random streams consumed in the same order, so this module uses the same three sources the
reference uses — `numpy.random` (legacy global state) for the beta draws, networkx's `stochastic_block_model` for the
edges, Python's `random` for everything planted on the graph — and documents the order below.  networkx is imported
lazily: only this generator needs it.  Pinned against blobs flattened from the reference's own objects for fixed seeds
(tests/golden/topology_random_s*.bin).

The reference's `new_environment` passes seed=None (a fresh network per process); `new_environment(..., seed=S)` here
additionally offers the reproducible variant used by the fixtures: `seed` goes to the traffic graph and `random.seed(seed)`
precedes the planting.
"""
from __future__ import annotations

import random
from collections import defaultdict
from typing import Dict, List, Optional

import numpy as np

from .. import model as m

ENV_IDENTIFIERS = m.Identifiers(
    properties=["breach_node"],
    ports=["SMB", "HTTP", "RDP"],
    local_vulnerabilities=["ScanWindowsCredentialManagerForRDP", "ScanWindowsExplorerRecentFiles", "ScanWindowsCredentialManagerForSMB"],
    remote_vulnerabilities=["Traceroute"],
)


def generate_random_traffic_network(n_clients: int = 200, n_servers: Optional[Dict[str, int]] = None, seed: Optional[int] = 0,
                                    tolerance=np.float32(1e-3),
                                    alpha=np.array([(0.1, 0.3), (0.18, 0.09)], dtype=float),
                                    beta=np.array([(100, 10), (10, 100)], dtype=float)):
    """Directed graph over integer node ids whose edges carry the set of protocols seen between the two machines
    (generate_network.py:23-76).  Per protocol, in dictionary order: re-seed numpy, draw the 2x2 edge probabilities,
    scale (SMB x3, RDP x4), clip to [tolerance, 1 - tolerance], sample the block model with the same seed."""
    import networkx as nx

    if n_servers is None:
        n_servers = {"SMB": 1, "HTTP": 1, "RDP": 1}
    scale = {"SMB": 3, "RDP": 4}
    protocols_of_edge = defaultdict(set)
    for protocol, count in n_servers.items():
        np.random.seed(seed)
        p = np.random.beta(a=alpha, b=beta, size=(2, 2))
        if protocol in scale:
            p = scale[protocol] * p
        p = np.clip(p, a_min=tolerance, a_max=np.float32(1.0 - tolerance))
        blocks = nx.stochastic_block_model(sizes=[n_clients, count], p=p, directed=True, seed=seed)
        for edge in blocks.edges:
            protocols_of_edge[edge].add(protocol)
    graph = nx.DiGraph()
    for (u, v), protocols in list(protocols_of_edge.items()):
        graph.add_edge(u, v, protocol=protocols)
    return graph


class _Planter:
    """State of cyberbattle_model_from_traffic_graph (generate_network.py:79-263): the password counter, the valid passwords
    assigned per (node, port) — services hold these very list objects, so passwords assigned while LATER nodes get their
    vulnerabilities still open the service — and the probabilities."""

    def __init__(self, edges, prob):
        self.edges = edges                      # [(source, target, protocols)] in graph.edges() order, string ids
        self.prob = prob
        self.n_passwords = 0
        self.valid: Dict[tuple, List[str]] = defaultdict(list)

    def new_password(self) -> str:
        self.n_passwords += 1
        return f"unique_pwd{self.n_passwords}"

    def new_valid_password(self, node, port) -> str:
        pwd = self.new_password()
        self.valid[node, port].append(pwd)
        return pwd

    def cached_credential(self, node, port) -> str:
        # draw 1: the cached password was rotated since (an invalid one); draw 2: another machine's password is reused
        if random.random() < self.prob["changed"]:
            return self.new_password()
        if random.random() < self.prob["shared"]:
            if (node, port) not in self.valid:
                return self.new_valid_password(node, port)
            return random.choice(self.valid[node, port])
        return self.new_valid_password(node, port)

    def targets(self, source, protocol) -> List[str]:
        return [t for (s, t, protocols) in self.edges if s == source and protocol in protocols]

    def vulnerabilities(self, node) -> m.VulnerabilityLibrary:
        """Order of draws per node: RDP credentials (keep-draw, then the credential's own draws, per neighbour), SMB recent
        files (one draw per neighbour), SMB credentials, traceroute (one draw per SMB neighbour; the reference's
        `smb_neighbors or rdp_neighbors` is the SMB list whenever the vulnerability exists)."""
        lib: m.VulnerabilityLibrary = {}
        rdp, smb = self.targets(node, "RDP"), self.targets(node, "SMB")
        L, R = m.VulnerabilityType.LOCAL, m.VulnerabilityType.REMOTE

        def leaked(neighbours, port, keep):
            out = []
            for t in neighbours:
                if random.random() < keep:
                    out.append(m.CachedCredential(node=t, port=port, credential=self.cached_credential(t, port)))
            return m.LeakedCredentials(credentials=out)

        if rdp:
            lib["ScanWindowsCredentialManagerForRDP"] = m.VulnerabilityInfo(
                description="", type=L, outcome=leaked(rdp, "RDP", self.prob["rdp"]), cost=2.0)
        if smb:
            lib["ScanWindowsExplorerRecentFiles"] = m.VulnerabilityInfo(
                description="", type=L, outcome=m.LeakedNodesId([t for t in smb if random.random() < self.prob["shares"]]), cost=1.0)
            lib["ScanWindowsCredentialManagerForSMB"] = m.VulnerabilityInfo(
                description="", type=L, outcome=leaked(smb, "SMB", self.prob["smb"]), cost=2.0)
        if smb and rdp:
            lib["Traceroute"] = m.VulnerabilityInfo(
                description="", type=R, outcome=m.LeakedNodesId([t for t in smb if random.random() < self.prob["traceroute"]]), cost=5.0)
        return lib


def cyberbattle_model_from_traffic_graph(traffic_graph, cached_smb_password_probability=0.75, cached_rdp_password_probability=0.8,
                                         cached_accessed_network_shares_probability=0.6,
                                         cached_password_has_changed_probability=0.1, traceroute_discovery_probability=0.5,
                                         probability_two_nodes_use_same_password_to_access_given_resource=0.8) -> m.Network:
    """Plant services, values and leak vulnerabilities on a traffic graph.  Sequence (each step consumes `random`):
    entry node index; the entry node's vulnerabilities; for every other node in graph order a value in [0, 100] (its
    services are whatever passwords target it SO FAR, by reference to the live lists); then every other node's
    vulnerabilities.  One firewall configuration object is shared by all nodes."""
    node_ids = [str(n) for n in traffic_graph.nodes]
    edges = [(str(s), str(t), traffic_graph.edges[(s, t)]["protocol"]) for (s, t) in traffic_graph.edges()]
    planter = _Planter(edges, dict(smb=cached_smb_password_probability, rdp=cached_rdp_password_probability,
                                   shares=cached_accessed_network_shares_probability,
                                   changed=cached_password_has_changed_probability, traceroute=traceroute_discovery_probability,
                                   shared=probability_two_nodes_use_same_password_to_access_given_resource))
    allow = m.RulePermission.ALLOW
    firewall = m.FirewallConfiguration([m.FirewallRule("RDP", allow), m.FirewallRule("SMB", allow)],
                                       [m.FirewallRule("RDP", allow), m.FirewallRule("SMB", allow)])
    entry = node_ids[random.randrange(len(node_ids))]
    data: Dict[str, m.NodeInfo] = {
        entry: m.NodeInfo(services=[], value=0, properties=["breach_node"], vulnerabilities=planter.vulnerabilities(entry),
                          agent_installed=True, firewall=firewall, reimagable=False)}
    for node in node_ids:
        if node != entry:
            services = [m.ListeningService(name=port, allowedCredentials=planter.valid[target, port])
                        for (target, port) in list(planter.valid.keys()) if target == node]
            data[node] = m.NodeInfo(services=services, value=random.randint(0, 100), agent_installed=False, firewall=firewall)
    for node in node_ids:
        if node != entry:
            data[node].vulnerabilities = planter.vulnerabilities(node)
    return m.create_network({n: data[n] for n in node_ids})


def new_environment(n_servers_per_protocol: int, seed: Optional[int] = None) -> m.Environment:
    """generate_network.py:266-294 (50 clients, n servers per protocol, its probabilities); `seed` see the module docstring."""
    traffic = generate_random_traffic_network(
        seed=seed, n_clients=50,
        n_servers={"SMB": n_servers_per_protocol, "HTTP": n_servers_per_protocol, "RDP": n_servers_per_protocol},
        alpha=np.array([(1, 1), (0.2, 0.5)], dtype=float), beta=np.array([(1000, 10), (10, 100)], dtype=float))
    if seed is not None:
        random.seed(seed)
    network = cyberbattle_model_from_traffic_graph(
        traffic, cached_rdp_password_probability=0.8, cached_smb_password_probability=0.7,
        cached_accessed_network_shares_probability=0.8, cached_password_has_changed_probability=0.01,
        probability_two_nodes_use_same_password_to_access_given_resource=0.9)
    return m.Environment(network=network, vulnerability_library={}, identifiers=ENV_IDENTIFIERS)
