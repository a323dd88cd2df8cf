def parse_fasta_bytes(data):
    name, chunks = None, []
    for line in data.splitlines():
        if line.startswith(b">"):
            if name is not None:
                yield name, b"".join(chunks)
            name, chunks = line[1:].split()[0].decode(), []
        elif name is not None:
            chunks.append(line.strip())
    if name is not None:
        yield name, b"".join(chunks)
