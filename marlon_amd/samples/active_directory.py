"""Active-Directory style topologies (`ActiveDirectory-v0..9`, `ActiveDirectoryTiny-v0`).

Restated from src/CyberBattleSim/cyberbattle/samples/active_directory/generate_ad.py (identifiers :9-19, network
:22-151, seeded sizes :154-165) and tiny_ad.py (:6-112).  The generator draws from Python's `random` module in a fixed
order; the same seed therefore yields the reference's network: seed -> (clients, servers, users), then for the breach
node the size of the spoofing credential set and its members, then one draw per further workstation deciding whether its
users are admins.  Pinned against blobs flattened from the reference's own objects (tests/golden/topology_ad*.bin).

Shapes that matter for the engine: one firewall configuration OBJECT shared by every node (its two rule lists are
aliased across the whole network, see DESIGN.md "rule lists"); `DumpNTDS` leaks users x (servers + clients) credentials in
one action, so `maximum_discoverable_credentials_per_action` / `maximum_total_credentials` must be sized for it (the
registered envs use 50 000, which makes the reference's observation arrays enormous; sizes that just fit are enough);
`ScanForCreds` carries a success rate the step never consults (rates are unused on the hot path).
"""
from __future__ import annotations

import random
from typing import Dict

from .. import model as m

ENV_IDENTIFIERS = m.Identifiers(
    properties=["breach_node", "domain_controller", "admin"],
    ports=["SMB", "AD", "SHELL"],
    local_vulnerabilities=["FindDomainControllers", "EnumerateFileShares", "AuthorizationSpoofAndCrack", "ScanForCreds",
                           "DumpNTDS", "ProbeAdmin"],
    remote_vulnerabilities=["PasswordSpray"],
)

_L = m.VulnerabilityType.LOCAL


def _firewall() -> m.FirewallConfiguration:
    def allow_all():
        return [m.FirewallRule(p, m.RulePermission.ALLOW) for p in ("SMB", "AD", "SHELL")]
    return m.FirewallConfiguration(allow_all(), allow_all())


def _cred(node, port, credential) -> m.CachedCredential:
    return m.CachedCredential(node=node, port=port, credential=credential)


class _Library:
    """The vulnerability dictionaries of the four machine roles; insertion order is part of the data (it decides nothing
    on the step path, but the blob lists a node's vulnerabilities in it)."""

    def __init__(self, shares_outcome, with_rates: bool):
        self.shares_outcome = shares_outcome
        self.with_rates = with_rates

    def base(self) -> m.VulnerabilityLibrary:
        scan = dict(description="", type=_L, precondition=m.Precondition("admin"),
                    outcome=m.LeakedCredentials(credentials=[_cred("domain_controller_1", "AD", "dc_1")]))
        if self.with_rates:
            scan["rates"] = m.Rates(successRate=0.9)
        return {
            "FindDomainControllers": m.VulnerabilityInfo(description="", type=_L, outcome=m.LeakedNodesId(nodes=["domain_controller_1"])),
            "EnumerateFileShares": m.VulnerabilityInfo(description="", type=_L, outcome=self.shares_outcome()),
            "ProbeAdmin": m.VulnerabilityInfo(description="", type=_L, outcome=m.ProbeFailed()),
            "ScanForCreds": m.VulnerabilityInfo(**scan),
        }

    def admin(self) -> m.VulnerabilityLibrary:
        lib = self.base()
        lib["ProbeAdmin"] = m.VulnerabilityInfo(description="", type=_L, outcome=m.ProbeSucceeded(discovered_properties=["admin"]))
        return lib

    def breach(self, leaked) -> m.VulnerabilityLibrary:
        lib = self.base()
        lib["AuthorizationSpoofAndCrack"] = m.VulnerabilityInfo(description="", type=_L, outcome=m.LeakedCredentials(credentials=leaked))
        return lib

    def controller(self, leaked) -> m.VulnerabilityLibrary:
        lib = self.base()
        lib["DumpNTDS"] = m.VulnerabilityInfo(description="", type=_L, precondition=m.Precondition("domain_controller"),
                                              outcome=m.LeakedCredentials(leaked))
        return lib


def create_network_from_smb_traffic(n_clients: int, n_servers: int, n_users: int) -> m.Network:
    """generate_ad.py:22-151.  Node order: workstations, shares, the domain controller."""
    fw = _firewall()
    shares = [f"share_{i}" for i in range(n_servers)]
    lib = _Library(lambda: m.LeakedNodesId(nodes=list(shares)), with_rates=True)

    # breach node first: its two `random` consumers come before the per-workstation admin draws
    spoofed = set(random.randrange(0, n_users) for _ in range(random.randrange(3, n_clients)))
    breach_leak = [_cred(s, "SMB", f"user_{u}") for u in spoofed for s in shares] + \
                  [_cred(f"workstation_{u % n_clients}", "SHELL", f"user_{u}") for u in spoofed]
    nodes: Dict[m.NodeID, m.NodeInfo] = {
        "workstation_0": m.NodeInfo(services=[], value=0, properties=["breach_node"], vulnerabilities=lib.breach(breach_leak),
                                    agent_installed=True, firewall=fw, reimagable=False)}
    for i in range(1, n_clients):
        is_admin = random.random() > 0.2
        nodes[f"workstation_{i}"] = m.NodeInfo(
            services=[m.ListeningService(name="SHELL", allowedCredentials=[f"user_{u}" for u in range(n_users) if u % n_clients == i])],
            properties=["admin"] if is_admin else [], value=1, firewall=fw,
            vulnerabilities=lib.admin() if is_admin else lib.base())
    for i, share in enumerate(shares):
        nodes[share] = m.NodeInfo(
            services=[m.ListeningService(name="SMB", allowedCredentials=[f"user_{u}" for u in range(n_users) if u % n_servers == i])],
            properties=[], value=5, firewall=fw, vulnerabilities=lib.base())
    everything = [_cred(s, "SMB", f"user_{u}") for u in range(n_users) for s in shares] + \
                 [_cred(f"workstation_{w}", "SHELL", f"user_{u}") for w in range(n_clients) for u in range(n_users)]
    nodes["domain_controller_1"] = m.NodeInfo(
        services=[m.ListeningService(name="AD", allowedCredentials=["dc_1"])], properties=["domain_controller"], value=1000,
        firewall=fw, vulnerabilities=lib.controller(everything))
    return m.create_network(nodes)


def new_random_environment(seed) -> m.Environment:
    """generate_ad.py:154-165: 5-9 workstations, one share, 20-99 users, all from `random.seed(seed)`."""
    random.seed(seed)
    clients = random.randrange(5, 10)
    servers = random.randrange(1, 2)
    users = random.randrange(20, 100)
    return m.Environment(network=create_network_from_smb_traffic(clients, servers, users), vulnerability_library={},
                         identifiers=ENV_IDENTIFIERS)


def new_tiny_environment() -> m.Environment:
    """tiny_ad.py: a domain controller, the breach workstation and one admin workstation."""
    fw = _firewall()
    lib = _Library(m.ExploitFailed, with_rates=False)
    users = [f"user_{u}" for u in range(20)]
    nodes = {
        "domain_controller_1": m.NodeInfo(
            services=[m.ListeningService(name="AD", allowedCredentials=["dc_1"])], properties=["domain_controller"], value=100,
            firewall=fw, vulnerabilities=lib.controller([_cred("workstation_0", "SHELL", u) for u in users])),
        "workstation_0": m.NodeInfo(
            services=[m.ListeningService(name="SHELL", allowedCredentials=list(users))], value=0, properties=["breach_node"],
            vulnerabilities=lib.breach([_cred("workstation_1", "SHELL", "user_1")]), agent_installed=True, firewall=fw,
            reimagable=False),
        "workstation_1": m.NodeInfo(
            services=[m.ListeningService(name="SHELL", allowedCredentials=list(users))], properties=["admin"], value=1,
            firewall=fw, vulnerabilities=lib.admin()),
    }
    return m.Environment(network=m.create_network(nodes), vulnerability_library={}, identifiers=ENV_IDENTIFIERS)
