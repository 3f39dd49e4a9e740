"""Three-node debugging topology (`CyberBattleTiny-v0`).

Same network as the reference table (src/CyberBattleSim/cyberbattle/samples/toyctf/tinytoy.py:10-62; identifiers inferred
from the nodes, :67; new_environment :70-71), written as a spec table like samples/toy_ctf.py: a client finds the website in
its browser history, the website's page source reveals a directory, the directory leaks the website's SSH credential.  The
leaked credential names port "MySQL" while the website listens on "SSH" (the reference's data, kept): connecting with it on
SSH works because authorisation looks at the credential string, not at the port it was leaked for.  The website's incoming
rule list is the ONE `default_allow_rules` object; its outgoing list is a new list built from it.  Pinned against the blob
flattened from the reference's objects (tests/golden/topology_tiny.bin).
"""
from __future__ import annotations

from typing import Dict

from .. import model as m

# node -> (value, properties, services [(port, credentials)], {vulnerability: (type, outcome)}, flags)
_NODES = (
    ("Website", 1000, ("MySql", "Ubuntu", "nginx/1.10.3"), (("SSH", ("ReusedMySqlCred-web",)),),
     {"ScanPageSource": ("remote", ("nodes", "Website.Directory"))}, dict(owned="FLAG: Login using insecure SSH user/password", rules=True)),
    ("Website.Directory", 50, ("Ubuntu", "nginx/1.10.3", "CTFFLAG:Readme.txt-Discover secret data"), (("HTTPS", ()),),
     {"NavigateWebDirectoryFurther": ("remote", ("cred", "Website", "MySQL", "ReusedMySqlCred-web"))}, dict()),
    ("client", 0, ("CLIENT:Win10",), (),
     {"SearchEdgeHistory": ("local", ("nodes", "Website"))}, dict(installed=True, reimagable=False)),
)


def _build() -> Dict[m.NodeID, m.NodeInfo]:
    allow = m.RulePermission.ALLOW
    default_allow_rules = [m.FirewallRule("SSH", allow)]
    out: Dict[m.NodeID, m.NodeInfo] = {}
    for name, value, props, services, vulns, flags in _NODES:
        table = {}
        for vid, (kind, spec) in vulns.items():
            outcome = m.LeakedNodesId([spec[1]]) if spec[0] == "nodes" else \
                m.LeakedCredentials([m.CachedCredential(node=spec[1], port=spec[2], credential=spec[3])])
            table[vid] = m.VulnerabilityInfo(description="", outcome=outcome, cost=1.0,
                                             type=m.VulnerabilityType.LOCAL if kind == "local" else m.VulnerabilityType.REMOTE)
        kw = dict(services=[m.ListeningService(p, allowedCredentials=list(c)) for p, c in services], value=value,
                  properties=list(props), vulnerabilities=table, owned_string=flags.get("owned", ""),
                  agent_installed=flags.get("installed", False), reimagable=flags.get("reimagable", True))
        if flags.get("rules"):
            kw["firewall"] = m.FirewallConfiguration(
                incoming=default_allow_rules,
                outgoing=default_allow_rules + [m.FirewallRule("su", allow), m.FirewallRule("sudo", allow)])
        out[name] = m.NodeInfo(**kw)
    return out


nodes = _build()
global_vulnerability_library: Dict[m.VulnerabilityID, m.VulnerabilityInfo] = {}
ENV_IDENTIFIERS = m.infer_constants_from_nodes(list(nodes.items()), global_vulnerability_library)


def new_environment() -> m.Environment:
    return m.Environment(network=m.create_network(_build()), vulnerability_library=global_vulnerability_library,
                         identifiers=ENV_IDENTIFIERS)
