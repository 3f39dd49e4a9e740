"""Chain topology:  start -> (Linux -> Windows)* -> Linux[flag].

Produces the same network as the reference generator
(src/CyberBattleSim/cyberbattle/samples/chainpattern/chainpattern.py:56-76 identifiers,
:94-195 one Linux/Windows link, :198-239 whole chain, :242-243 new_environment): same node
insertion order (start, final Linux node, then links 1,3,5,...), same vulnerabilities, costs,
outcomes, credentials and firewall rules.  Written table-driven: each operating system is a
row of (vulnerability id -> spec) and the link index fills in the neighbour names.
tests/test_topology.py checks the flattened blob against the blob flattened from the
reference's own objects (fixture tests/golden/topology_*.bin).
"""
from __future__ import annotations

from typing import Dict

from .. import model as m

ENV_IDENTIFIERS = m.Identifiers(
    properties=["Windows", "Linux", "ApacheWebSite", "IIS_2019", "IIS_2020_patched", "MySql", "Ubuntu",
                "nginx/1.10.3", "SMB_vuln", "SMB_vuln_patched", "SQLServer", "Win10", "Win10Patched", "FLAG:Linux"],
    ports=["HTTPS", "GIT", "SSH", "RDP", "PING", "MySQL", "SSH-key", "su"],
    local_vulnerabilities=["ScanBashHistory", "ScanExplorerRecentFiles", "SudoAttempt", "CrackKeepPassX", "CrackKeepPass"],
    remote_vulnerabilities=["ProbeLinux", "ProbeWindows"],
)

LINUX_PROPS = ["MySql", "Ubuntu", "nginx/1.10.3"]
WINDOWS_PROPS = ["Windows", "Win10", "Win10Patched"]


def prefix(x: int, name: str) -> str:
    return f"{x}_{name}"


def rdp_password(index) -> str:
    return f"WindowsPassword!{index}"


def ssh_password(index) -> str:
    return f"LinuxPassword!{index}"


def _allow_all():
    return [m.FirewallRule(p, m.RulePermission.ALLOW) for p in ("RDP", "SSH", "HTTPS", "HTTP")]


def _vuln(kind: m.VulnerabilityType, outcome: m.VulnerabilityOutcome, cost: float, text: str = "", why: str = "") -> m.VulnerabilityInfo:
    return m.VulnerabilityInfo(description=text, type=kind, outcome=outcome, cost=cost, reward_string=why)


def _linux_node(n: int, shared_rules) -> m.NodeInfo:
    nxt = prefix(n + 1, "WindowsNode")
    L, R = m.VulnerabilityType.LOCAL, m.VulnerabilityType.REMOTE
    vulns = {
        "ProbeLinux": _vuln(R, m.ProbeSucceeded(["Ubuntu"]), 5.0, "probe: linux?"),
        "ProbeWindows": _vuln(R, m.ProbeFailed(), 5.0, "probe: windows?"),
        "ScanBashHistory": _vuln(L, m.LeakedNodesId([nxt]), 1.0, "bash history names the next machine"),
        "ScanExplorerRecentFiles": _vuln(L, m.ExploitFailed(), 10.0, "trap"),
        "SudoAttempt": _vuln(L, m.ExploitFailed(), 100.0, "trap"),
        "CrackKeepPassX": _vuln(L, m.LeakedCredentials([m.CachedCredential(nxt, "RDP", rdp_password(n + 1))]), 1.0,
                                "password vault holds the RDP password of the next machine"),
    }
    return m.NodeInfo(
        services=[m.ListeningService("HTTPS"), m.ListeningService("SSH", allowedCredentials=[ssh_password(n)])],
        # one rule list object serves both directions of every Linux link node, as in the reference (chainpattern.py:49-54,103):
        # an edit through one node is seen by all of them
        firewall=m.FirewallConfiguration(incoming=shared_rules, outgoing=shared_rules),
        value=100, properties=list(LINUX_PROPS), vulnerabilities=vulns,
        owned_string="Intermediate chain node owned, no intrinsic value")


def _windows_node(n: int) -> m.NodeInfo:
    """Windows node number n+1 of link n; it points at Linux node n+2."""
    nxt = prefix(n + 2, "LinuxNode")
    L, R = m.VulnerabilityType.LOCAL, m.VulnerabilityType.REMOTE
    vulns = {
        "ProbeLinux": _vuln(R, m.ProbeFailed(), 1.0, "probe: linux?"),
        "ProbeWindows": _vuln(R, m.ProbeSucceeded(["Windows"]), 1.0, "probe: windows?"),
        "ScanBashHistory": _vuln(L, m.ExploitFailed(), 100.0, "trap"),
        "ScanExplorerRecentFiles": _vuln(L, m.LeakedNodesId([nxt]), 1.0, "recent files name the next machine"),
        "SudoAttempt": _vuln(L, m.ExploitFailed(), 100.0, "trap"),
        "CrackKeepPassX": _vuln(L, m.ExploitFailed(), 100.0, "trap"),
        "CrackKeepPass": _vuln(L, m.LeakedCredentials([m.CachedCredential(nxt, "SSH", ssh_password(n + 2))]), 1.0,
                               "password vault holds the SSH password of the next machine"),
    }
    return m.NodeInfo(
        services=[m.ListeningService("HTTPS"), m.ListeningService("RDP", allowedCredentials=[rdp_password(n + 1)])],
        value=100, properties=list(WINDOWS_PROPS), vulnerabilities=vulns)


def create_network_chain_link(n: int, shared_rules=None) -> Dict[m.NodeID, m.NodeInfo]:
    shared_rules = _allow_all() if shared_rules is None else shared_rules
    return {prefix(n, "LinuxNode"): _linux_node(n, shared_rules), prefix(n + 1, "WindowsNode"): _windows_node(n)}


def create_chain_network(size: int) -> Dict[m.NodeID, m.NodeInfo]:
    if size % 2 == 1:
        raise ValueError(f"Chain size must be even: {size}")
    last = size + 1
    first_hop = m.CachedCredential(prefix(1, "LinuxNode"), "SSH", ssh_password(1))
    nodes: Dict[m.NodeID, m.NodeInfo] = {
        "start": m.NodeInfo(
            services=[], value=0, agent_installed=True, reimagable=False,
            vulnerabilities={"ScanExplorerRecentFiles": _vuln(m.VulnerabilityType.LOCAL, m.LeakedCredentials([first_hop]), 1.0,
                                                               "recent files hold the SSH password of machine 1")}),
        prefix(last, "LinuxNode"): m.NodeInfo(
            services=[m.ListeningService("HTTPS"), m.ListeningService("SSH", allowedCredentials=[ssh_password(last)])],
            value=1000, owned_string="FLAG: flag discovered!",
            properties=LINUX_PROPS + ["FLAG:Linux"], vulnerabilities={}),
    }
    shared = _allow_all()
    for i in range(1, size, 2):
        nodes.update(create_network_chain_link(i, shared))
    return nodes


def new_environment(size) -> m.Environment:
    return m.Environment(network=m.create_network(create_chain_network(size)), vulnerability_library={},
                         identifiers=ENV_IDENTIFIERS)
