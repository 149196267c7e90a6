"""A small populated LDBC SNB interactive database and five statements shaped like the reference's
friends-of-friends queries (benchmark/ldbc/queries/interactive-complex-3/5/6/9/11.sql: a derived table that
UNIONs the friends of one person with the friends of those friends, joined with person / place / message /
forum / organisation / tag columns, aggregated, ordered, limited).

The statements are generated here (table and column names are the schema's, benchmark/ldbc/schema.sql); the shipped
texts themselves are read from the reference tree where it exists (tests/test_plan_rule.py) and never stored.
"""
import numpy as np

PERSON_A = 6597069767251    # the person constants the shipped texts use
PERSON_B = 21990232556256
PERSON_X = 15393162789164
COUNTRIES = ["United_States", "Canada", "Germany", "France", "India", "China", "Brazil", "Kenya"]
TAGS = ["Hamid_Karzai", "Augustine_of_Hippo", "Napoleon", "Genghis_Khan", "Mozart", "Plato", "Rumi", "Pele",
        "Ada_Lovelace", "Hypatia", "Sappho", "Euclid"]

DDL = [
    "create table person (p_creationdate timestamp not null, p_personid bigint not null, p_firstname varchar not null, "
    "p_lastname varchar not null, p_gender varchar not null, p_birthday date not null, p_locationip varchar not null, "
    "p_browserused varchar not null, p_placeid bigint)",
    "create table knows (k_creationdate timestamp not null, k_person1id bigint not null, k_person2id bigint not null)",
    "create table place (pl_placeid bigint not null, pl_name varchar not null, pl_url varchar not null, "
    "pl_type varchar not null, pl_containerplaceid bigint)",
    "create table post (m_creationdate timestamp not null, m_messageid bigint not null, m_ps_imagefile varchar, "
    "m_locationip varchar not null, m_browserused varchar not null, m_ps_language varchar, m_content text, "
    "m_length int not null, m_creatorid bigint, m_ps_forumid bigint, m_locationid bigint)",
    "create table comment (m_creationdate timestamp not null, m_messageid bigint not null, m_locationip varchar not null, "
    "m_browserused varchar not null, m_content text not null, m_length int not null, m_creatorid bigint, "
    "m_locationid bigint, m_c_parentpostid bigint, m_c_parentcommentid bigint)",
    "create view message as select m_creationdate, m_messageid, m_ps_imagefile, m_locationip, m_browserused, m_content, "
    "m_length, m_creatorid, m_ps_forumid, m_locationid, null as m_c_replyof from post union all "
    "select m_creationdate, m_messageid, null as m_ps_imagefile, m_locationip, m_browserused, m_content, m_length, "
    "m_creatorid, null as m_ps_forumid, m_locationid, coalesce(m_c_parentpostid, m_c_parentcommentid) m_c_replyof from comment",
    "create table forum (f_creationdate timestamp not null, f_forumid bigint not null, f_title varchar not null, "
    "f_moderatorid bigint)",
    "create table forum_person (fp_creationdate timestamp not null, fp_forumid bigint not null, fp_personid bigint not null)",
    "create table organisation (o_organisationid bigint not null, o_type varchar not null, o_name varchar not null, "
    "o_url varchar not null, o_placeid bigint)",
    "create table person_company (pc_creationdate timestamp not null, pc_personid bigint not null, "
    "pc_organisationid bigint not null, pc_workfrom int not null)",
    "create table tag (t_tagid bigint not null, t_name varchar not null, t_url varchar not null, t_tagclassid bigint not null)",
    "create table message_tag (mt_creationdate timestamp not null, mt_messageid bigint not null, mt_tagid bigint not null)",
]


def _ts(rng, n, lo="2010-01-01", hi="2012-12-01"):
    a, b = np.datetime64(lo, "s").astype(np.int64), np.datetime64(hi, "s").astype(np.int64)
    return [str(np.datetime64(int(t), "s")).replace("T", " ") for t in rng.integers(a, b, n)]


def _insert(d, table, rows, batch=400):
    def lit(v):
        if v is None:
            return "NULL"
        if isinstance(v, str):
            return "'" + v.replace("'", "''") + "'"
        return str(int(v))
    for i in range(0, len(rows), batch):
        d.execute(f"INSERT INTO {table} VALUES " + ", ".join("(" + ", ".join(lit(v) for v in r) + ")" for r in rows[i:i + batch]))


def populate(d, seed=11, n_person=260, create=True):
    """Fill the tables the five statements read.  Returns the person ids."""
    rng = np.random.default_rng(seed)
    if create:
        for stmt in DDL:
            d.execute(stmt)
    # places: countries 1.., cities 100.. inside them
    countries = [(1 + i, name, "u", "country", None) for i, name in enumerate(COUNTRIES)]
    cities = [(100 + i, f"city{i}", "u", "city", 1 + i % len(COUNTRIES)) for i in range(40)]
    _insert(d, "place", countries + cities)
    ids = np.unique(np.concatenate([[PERSON_A, PERSON_B, PERSON_X, 19791209310731],
                                    rng.integers(1, 2**44, n_person).astype(np.int64)]))
    ts = _ts(rng, ids.size)
    _insert(d, "person", [(ts[i], int(p), f"first{i % 37}", f"last{i % 53}", "f" if i % 2 else "m", "1990-01-01", "ip", "br",
                           100 + int(rng.integers(0, 40))) for i, p in enumerate(ids)])
    # knows: symmetric, the constant persons well connected
    a = ids[rng.integers(0, ids.size, 2600)]
    b = ids[rng.integers(0, ids.size, 2600)]
    hub = np.concatenate([np.full(25, PERSON_A), np.full(25, PERSON_B), np.full(6, PERSON_X)])
    a, b = np.concatenate([a, hub]), np.concatenate([b, ids[rng.integers(0, ids.size, hub.size)]])
    keep = a != b
    a, b = a[keep], b[keep]
    pairs = np.unique(np.stack([np.concatenate([a, b]), np.concatenate([b, a])], 1), axis=0)
    kts = _ts(rng, pairs.shape[0])
    _insert(d, "knows", [(kts[i], int(p[0]), int(p[1])) for i, p in enumerate(pairs)])
    # forums, memberships
    fts = _ts(rng, 30, "2010-01-01", "2011-06-01")
    _insert(d, "forum", [(fts[i], 1000 + i, f"forum {i}", int(ids[rng.integers(0, ids.size)])) for i in range(30)])
    mts = _ts(rng, 2500, "2011-01-01", "2012-06-01")
    _insert(d, "forum_person", [(mts[i], 1000 + int(rng.integers(0, 30)), int(ids[rng.integers(0, ids.size)])) for i in range(2500)])
    # posts and comments, located in countries
    n_post, n_com = 3000, 1500
    pts = _ts(rng, n_post)
    _insert(d, "post", [(pts[i], 10_000 + i, None if i % 3 else f"img{i}.png", "ip", "br", "en", f"post body {i}", 10 + i % 90,
                         int(ids[rng.integers(0, ids.size)]), 1000 + int(rng.integers(0, 30)), 1 + int(rng.integers(0, len(COUNTRIES))))
                        for i in range(n_post)])
    cts = _ts(rng, n_com)
    _insert(d, "comment", [(cts[i], 50_000 + i, "ip", "br", f"comment body {i}", 5 + i % 40, int(ids[rng.integers(0, ids.size)]),
                            1 + int(rng.integers(0, len(COUNTRIES))), 10_000 + int(rng.integers(0, n_post)), None)
                           for i in range(n_com)])
    _insert(d, "tag", [(500 + i, name, "u", 1) for i, name in enumerate(TAGS)])
    tt = _ts(rng, 7000)
    mt = np.unique(np.stack([10_000 + rng.integers(0, n_post, 7000), 500 + rng.integers(0, len(TAGS), 7000)], 1), axis=0)
    _insert(d, "message_tag", [(tt[i], int(r[0]), int(r[1])) for i, r in enumerate(mt)])
    _insert(d, "organisation", [(700 + i, "company", f"org{i}", "u", 1 + i % len(COUNTRIES)) for i in range(24)])
    wts = _ts(rng, 500)
    _insert(d, "person_company", [(wts[i], int(ids[rng.integers(0, ids.size)]), 700 + int(rng.integers(0, 24)),
                                   2000 + int(rng.integers(0, 14))) for i in range(500)])
    return ids


def friends(person, excluded):
    """friends of `person` UNION friends of friends other than `excluded` (the derived table `f` of the five queries)"""
    return (f"(select k_person2id from knows where k_person1id = {person} union "
            f"select k2.k_person2id from knows k1, knows k2 where k1.k_person1id = {person} "
            f"and k1.k_person2id = k2.k_person1id and k2.k_person2id <> {excluded}) f")


def statements():
    """name -> SQL; every result is fully ordered (ties broken on keys) so that plans can be compared row by row"""
    def located(country, alias, until):
        return (f"(select m_creatorid as creator, count(*) as {alias} from message, place where m_locationid = pl_placeid "
                f"and pl_name = '{country}' and m_creationdate >= '2010-07-21T22:00:00' and m_creationdate < '{until}' "
                f"group by m_creatorid) {alias}s")
    s = {}
    # friends abroad who posted from two given countries (shape of interactive-complex-3)
    s["ic3"] = (f"select p_personid, p_firstname, p_lastname, ct1, ct2, ct1 + ct2 as total from {friends(PERSON_A, PERSON_X)}, "
                f"person, place p1, place p2, {located('United_States', 'ct1', '2012-07-26T22:00:00')}, "
                f"{located('Canada', 'ct2', '2012-01-26T22:00:00')} "
                "where f.k_person2id = p_personid and p_placeid = p1.pl_placeid and p1.pl_containerplaceid = p2.pl_placeid "
                "and p2.pl_name <> 'United_States' and p2.pl_name <> 'Canada' and f.k_person2id = ct1s.creator "
                "and ct1s.creator = ct2s.creator order by 6 desc, 1 limit 20")
    # forums the friends joined lately, by the posts those friends made there (interactive-complex-5)
    s["ic5"] = ("select f_title, count(m_messageid) from (select f_title, f_forumid, f.k_person2id from forum, forum_person, "
                f"{friends(PERSON_B, PERSON_B)} where f_forumid = fp_forumid and fp_personid = f.k_person2id "
                "and fp_creationdate >= '2011-07-21T22:00:00') tmp left join message on tmp.f_forumid = m_ps_forumid "
                "and m_creatorid = tmp.k_person2id group by f_forumid, f_title order by 2 desc, f_forumid limit 20")
    # tags that occur together with one tag on the friends' posts (interactive-complex-6)
    s["ic6"] = (f"select t_name, count(*) from tag, message_tag, message, {friends(PERSON_B, PERSON_B)} "
                "where m_creatorid = f.k_person2id and m_c_replyof is null and m_messageid = mt_messageid and mt_tagid = t_tagid "
                f"and t_name <> '{TAGS[0]}' and exists (select * from tag, message_tag where mt_messageid = m_messageid "
                f"and mt_tagid = t_tagid and t_name = '{TAGS[0]}') group by t_name order by 2 desc, t_name limit 10")
    # the friends' latest messages with their authors' names (interactive-complex-9)
    s["ic9"] = ("select p_personid, p_firstname, p_lastname, m_messageid, coalesce(m_ps_imagefile, '') || coalesce(m_content, '') "
                f"as content, m_creationdate from {friends(PERSON_B, PERSON_B)}, person, message where p_personid = m_creatorid "
                "and p_personid = f.k_person2id and m_creationdate < '2012-07-26T22:00:00' "
                "order by m_creationdate desc, m_messageid asc limit 20")
    # friends who work in one country since before a year (interactive-complex-11)
    s["ic11"] = ("select p_personid, p_firstname, p_lastname, o_name, pc_workfrom from person, person_company, organisation, place, "
                 f"{friends(PERSON_B, PERSON_B)} where p_personid = f.k_person2id and p_personid = pc_personid "
                 "and pc_organisationid = o_organisationid and pc_workfrom < 2012 and o_placeid = pl_placeid "
                 "and pl_name = 'United_States' order by pc_workfrom, p_personid, o_name desc limit 10")
    return s
