"""A small synthetic topology that touches every rule of the step at least once.

Not a reference topology: it exists so that the golden traces (tests/golden/sink_*.npz, captured by
running the REFERENCE on this very topology) cover the branches Chain and ToyCtf never reach:
a global vulnerability library, Admin/System privilege escalation with tag-dependent preconditions
(the actions_test.py:39-73 UACME family), a remote LateralMove exploit, CustomerData, ProbeFailed,
multi-entry LeakedCredentials / LeakedNodesId lists with repeats, BLOCK rules that shadow later ALLOW
rules, a stopped service, a credential accepted by two nodes, non-reimagable nodes and non-uniform
SLA weights (so the availability sum is order dependent).  `build(m)` takes the model module to
build with: marlon_amd.model here, the reference's simulation.model in oracle/refharness.
"""
from __future__ import annotations


def build(m, entry_reimagable: bool = False):
    A, B = m.RulePermission.ALLOW, m.RulePermission.BLOCK
    L, R = m.VulnerabilityType.LOCAL, m.VulnerabilityType.REMOTE
    admin, system = m.AdminEscalation().tag, m.SystemEscalation().tag

    def rules(*pairs):
        return [m.FirewallRule(p, perm) for p, perm in pairs]

    def vuln(kind, outcome, cost=1.0, pre=None):
        kw = dict(description="", type=kind, outcome=outcome, cost=cost)
        if pre is not None:
            kw["precondition"] = m.Precondition(pre)
        return m.VulnerabilityInfo(**kw)

    library = {
        "UacBypass": vuln(L, m.AdminEscalation(), 2.0, f"Windows&Win10&(~({admin}|{system}))"),
        "TokenSteal": vuln(L, m.SystemEscalation(), 3.0, f"Windows & {admin} & not {system}"),
        "DumpSecrets": vuln(L, m.LeakedCredentials([m.CachedCredential("db", "SQL", "sa-password"),
                                                     m.CachedCredential("files", "SMB", "shared-secret")]), 4.0,
                            f"Windows&({admin}|{system})"),
        "Ping": vuln(R, m.ProbeFailed(), 1.0),
        "BruteRdp": vuln(R, m.LateralMove(), 9.0, "Windows&PortRDPOpen"),
    }

    def node(**kw):
        kw.setdefault("services", [])
        return m.NodeInfo(**kw)

    nodes = {
        "entry": node(
            value=0, agent_installed=True, reimagable=entry_reimagable, properties=["Linux"],
            firewall=m.FirewallConfiguration(incoming=rules(("SSH", A)), outgoing=rules(("RDP", A), ("SSH", A), ("SQL", A), ("SMB", B), ("SMB", A), ("HTTP", A))),
            vulnerabilities={
                "Recon": vuln(L, m.LeakedNodesId(["ws1", "ws2", "ws1", "web"]), 1.0),
                "ReadNotes": vuln(L, m.LeakedCredentials([m.CachedCredential("ws1", "RDP", "alice"),
                                                          m.CachedCredential("ws2", "RDP", "alice"),
                                                          m.CachedCredential("ws1", "RDP", "alice")]), 2.0),
            }),
        "ws1": node(
            value=40, sla_weight=2.0, properties=["Windows", "Win10", "PortRDPOpen"],
            services=[m.ListeningService("RDP", allowedCredentials=["alice"]), m.ListeningService("HTTP")],
            vulnerabilities={
                "FingerprintOs": vuln(R, m.ProbeSucceeded(["Windows", "Win10"]), 2.0),
                "BrowserHistory": vuln(L, m.LeakedNodesId(["db", "files"]), 1.0),
            }),
        "ws2": node(
            value=35, sla_weight=0.3, properties=["Windows", "Win10"],
            services=[m.ListeningService("RDP", allowedCredentials=["alice", "bob"]),
                      m.ListeningService("RDP", allowedCredentials=["carol"], running=False)],
            firewall=m.FirewallConfiguration(incoming=rules(("RDP", A), ("SSH", B)), outgoing=rules(("SQL", A), ("RDP", A))),
            vulnerabilities={
                "FingerprintOs": vuln(R, m.ProbeSucceeded(["Win10"]), 2.0),
                "KeyFile": vuln(L, m.LeakedCredentials([m.CachedCredential("web", "SSH", "deploy-key")]), 1.0, f"{admin}|Linux"),
            }),
        "web": node(
            value=60, sla_weight=1.5, reimagable=False, properties=["Linux", "Nginx"],
            services=[m.ListeningService("SSH", allowedCredentials=["deploy-key"]), m.ListeningService("HTTP")],
            firewall=m.FirewallConfiguration(incoming=rules(("HTTP", A), ("SSH", A)), outgoing=rules(("SQL", A), ("SMB", A))),
            vulnerabilities={
                "ScrapeSite": vuln(R, m.LeakedCredentials([m.CachedCredential("db", "SQL", "web-app")]), 1.0, "Nginx&~Windows"),
                "SudoTrap": vuln(L, m.ExploitFailed(), 50.0),
            }),
        "db": node(
            value=200, sla_weight=0.7, properties=["Linux", "SqlServer"],
            services=[m.ListeningService("SQL", allowedCredentials=["sa-password", "web-app"])],
            firewall=m.FirewallConfiguration(incoming=rules(("SQL", A)), outgoing=[]),
            vulnerabilities={"ExportTables": vuln(R, m.CustomerData(), 6.0, "SqlServer")}),
        "files": node(
            value=80, properties=["Windows", "PortRDPOpen"],
            services=[m.ListeningService("SMB", allowedCredentials=["shared-secret"])],
            firewall=m.FirewallConfiguration(incoming=rules(("SMB", A), ("RDP", A)), outgoing=rules(("RDP", A))),
            vulnerabilities={}),
    }
    identifiers = m.Identifiers(
        properties=["Windows", "Linux", "Win10", "PortRDPOpen", "Nginx", "SqlServer"],
        ports=["RDP", "SSH", "HTTP", "SQL", "SMB"],
        local_vulnerabilities=["UacBypass", "TokenSteal", "DumpSecrets", "Recon", "ReadNotes", "BrowserHistory", "KeyFile", "SudoTrap"],
        remote_vulnerabilities=["Ping", "BruteRdp", "FingerprintOs", "ScrapeSite", "ExportTables"],
    )
    return m.Environment(network=m.create_network(nodes), vulnerability_library=library, identifiers=identifiers)


def new_environment():
    from .. import model
    return build(model)
