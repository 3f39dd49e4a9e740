"""Three-node debugging topology (`CyberBattleTiny-v0`).

Same network as the reference table (src/CyberBattleSim/cyberbattle/samples/toyctf/tinytoy.py:10-62; identifiers inferred
from the nodes, :67; new_environment :70-71): a client that finds the website in its browser history, the website whose
page source reveals a directory, the directory that leaks the website's SSH credential.  Note the leaked credential names
port "MySQL" while the website listens on "SSH" (the reference's data, kept): connecting with it on SSH works because
authorisation looks at the credential string, not at the port it was leaked for.  `default_allow_rules` is ONE list object
shared by the website's incoming rules and the head of its outgoing rules' construction (outgoing is a new list:
`default_allow_rules + [...]`).  Pinned against the blob flattened from the reference's objects
(tests/golden/topology_tiny.bin).
"""
from __future__ import annotations

from typing import Dict

from .. import model as m

_ALLOW = m.RulePermission.ALLOW


def _build() -> Dict[m.NodeID, m.NodeInfo]:
    default_allow_rules = [m.FirewallRule("SSH", _ALLOW)]

    def vuln(kind, outcome):
        return m.VulnerabilityInfo(description="", type=kind, outcome=outcome, cost=1.0)

    return {
        "Website": m.NodeInfo(
            services=[m.ListeningService("SSH", allowedCredentials=["ReusedMySqlCred-web"])],
            firewall=m.FirewallConfiguration(
                incoming=default_allow_rules,
                outgoing=default_allow_rules + [m.FirewallRule("su", _ALLOW), m.FirewallRule("sudo", _ALLOW)]),
            value=1000, properties=["MySql", "Ubuntu", "nginx/1.10.3"],
            owned_string="FLAG: Login using insecure SSH user/password",
            vulnerabilities=dict(ScanPageSource=vuln(m.VulnerabilityType.REMOTE, m.LeakedNodesId(["Website.Directory"])))),
        "Website.Directory": m.NodeInfo(
            services=[m.ListeningService("HTTPS")], value=50,
            properties=["Ubuntu", "nginx/1.10.3", "CTFFLAG:Readme.txt-Discover secret data"],
            vulnerabilities=dict(NavigateWebDirectoryFurther=vuln(
                m.VulnerabilityType.REMOTE,
                m.LeakedCredentials([m.CachedCredential(node="Website", port="MySQL", credential="ReusedMySqlCred-web")])))),
        "client": m.NodeInfo(
            services=[], properties=["CLIENT:Win10"], value=0,
            vulnerabilities=dict(SearchEdgeHistory=vuln(m.VulnerabilityType.LOCAL, m.LeakedNodesId(["Website"]))),
            agent_installed=True, reimagable=False),
    }


nodes = _build()
global_vulnerability_library: Dict[m.VulnerabilityID, m.VulnerabilityInfo] = {}
ENV_IDENTIFIERS = m.infer_constants_from_nodes(list(nodes.items()), global_vulnerability_library)


def new_environment() -> m.Environment:
    return m.Environment(network=m.create_network(_build()), vulnerability_library=global_vulnerability_library,
                         identifiers=ENV_IDENTIFIERS)
