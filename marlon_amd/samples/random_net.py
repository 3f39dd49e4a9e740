"""Random N-node topology for BASELINE.json config 5 (256 nodes, mixed vulnerability/firewall tables).

Modelled on what the reference's generator produces (simulation/generate_network.py:79-263: per-node
services with allowed-credential lists, ALLOW/BLOCK firewall rules, local vulnerabilities leaking
credentials / node ids, remote Traceroute-style vulnerabilities), but it is this build's own,
deterministic generator (numpy PCG64 from `seed`), not a restatement of that file: generator parity
is a "next" row.  `build(m, ...)` takes the model module so that the same
topology can be instantiated with the reference's classes for golden traces.
"""
from __future__ import annotations

import numpy as np

PORTS = ["RDP", "SSH", "HTTP", "HTTPS", "SMB", "SQL", "FTP", "WMI"]
PROPS = ["Windows", "Linux", "Win10", "Win7", "Ubuntu", "PortRDPOpen", "PortSSHOpen", "SqlServer", "WebServer",
         "FileShare", "DomainJoined", "Patched", "Backup", "Dev", "Prod", "Legacy"]


def build(m, n_nodes: int = 256, seed: int = 0, n_start: int = 1):
    rng = np.random.Generator(np.random.PCG64(seed))
    A, B = m.RulePermission.ALLOW, m.RulePermission.BLOCK
    L, R = m.VulnerabilityType.LOCAL, m.VulnerabilityType.REMOTE
    admin = m.AdminEscalation().tag
    names = [f"n{i:03d}" for i in range(n_nodes)]
    pwd = {i: f"pw-{i:03d}" for i in range(n_nodes)}

    def pick(k, lo=0):
        return [int(x) for x in rng.integers(lo, n_nodes, size=k)]

    def vuln(kind, outcome, cost, pre=None):
        kw = dict(description="", type=kind, outcome=outcome, cost=float(cost))
        if pre is not None:
            kw["precondition"] = m.Precondition(pre)
        return m.VulnerabilityInfo(**kw)

    library = {
        "Escalate": vuln(L, m.AdminEscalation(), 3, f"Windows&~{admin}"),
        "Traceroute": vuln(R, m.ProbeFailed(), 2),
    }
    nodes = {}
    for i, name in enumerate(names):
        is_win = bool(rng.integers(0, 2))
        props = ["Windows" if is_win else "Linux"] + [PROPS[int(j)] for j in sorted(set(rng.integers(2, len(PROPS), size=3)))]
        main_port = "RDP" if is_win else "SSH"
        svc_ports = [main_port] + [PORTS[int(j)] for j in sorted(set(rng.integers(2, len(PORTS), size=int(rng.integers(0, 3))))) if PORTS[int(j)] != main_port]
        services = [m.ListeningService(main_port, allowedCredentials=[pwd[i]] + ([pwd[pick(1)[0]]] if rng.random() < 0.2 else []))]
        services += [m.ListeningService(p) for p in svc_ports[1:]]
        fw_in = [m.FirewallRule(p, A) for p in svc_ports]
        if rng.random() < 0.15:
            fw_in.insert(0, m.FirewallRule(PORTS[int(rng.integers(0, len(PORTS)))], B))
        fw_out = [m.FirewallRule(p, B if rng.random() < 0.1 else A) for p in PORTS]
        vulns = {}
        nbrs = pick(int(rng.integers(1, 4)))
        vulns["ScanNeighbours"] = vuln(L, m.LeakedNodesId([names[j] for j in nbrs]), 1)
        tgt = pick(int(rng.integers(1, 3)))
        vulns["DumpCredentials"] = vuln(L, m.LeakedCredentials([m.CachedCredential(names[j], "SSH", pwd[j]) for j in tgt]), 2,
                                        "Windows|Linux&~Legacy" if rng.random() < 0.5 else None)
        if rng.random() < 0.5:
            vulns["ProbeOs"] = vuln(R, m.ProbeSucceeded([props[0]]), 1)
        if rng.random() < 0.1:
            vulns["StealData"] = vuln(R, m.CustomerData(), 5, "SqlServer|FileShare")
        nodes[name] = m.NodeInfo(
            services=services, vulnerabilities=vulns, value=int(rng.integers(0, 11)) * 10, properties=props,
            firewall=m.FirewallConfiguration(incoming=fw_in, outgoing=fw_out),
            agent_installed=(i < n_start), reimagable=(i >= n_start), sla_weight=1.0)
    # a credential's port must match what its node really listens on: patch RDP/SSH by the final OS of the target
    for name, info in nodes.items():
        v = info.vulnerabilities["DumpCredentials"]
        fixed = [m.CachedCredential(c.node, "RDP" if "Windows" in nodes[c.node].properties else "SSH", c.credential)
                 for c in v.outcome.credentials]
        info.vulnerabilities["DumpCredentials"] = v._replace(outcome=m.LeakedCredentials(fixed))
    identifiers = m.Identifiers(
        properties=list(PROPS), ports=list(PORTS),
        local_vulnerabilities=["Escalate", "ScanNeighbours", "DumpCredentials"],
        remote_vulnerabilities=["Traceroute", "ProbeOs", "StealData"])
    return m.Environment(network=m.create_network(nodes), vulnerability_library=library, identifiers=identifiers)


def new_environment(n_nodes: int = 256, seed: int = 0):
    from .. import model
    return build(model, n_nodes=n_nodes, seed=seed)
